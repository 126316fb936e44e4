set -e
python -m pytest tests/test_kernels_gpu.py -x -q 2>&1 | tail -3
echo "tune off"; SY11_TUNE=0 python tools/conv_sweep.py 2>/dev/null | tail -1
echo "tune on"; python tools/conv_sweep.py -v 2>/dev/null > gpurun_out/sw_tune.txt; tail -1 gpurun_out/sw_tune.txt
python bench.py --no-cpu-baseline > gpurun_out/bench_tune.json 2>gpurun_out/bench_tune.err; python -c "
import json; d=json.load(open('gpurun_out/bench_tune.json')); print(d['value'], d['ms_per_step']); print(d['roofline']['families_ms'])"
