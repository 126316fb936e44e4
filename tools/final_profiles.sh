# Everything profiles/<round>/ is regenerated from at the last kernel commit: the full -m gpu suite, one-queue and two-queue kernel traces,
# PMC traffic, the predict / val leg, a 300-step soak and the ordered-mode trace -> gpurun_out/r7f/ (copy what is to be judged into profiles/).
set -e
mkdir -p gpurun_out/r7f
timeout -k 10 900 python -m pytest tests -q -m gpu -x > gpurun_out/r7f/gpu_suite.txt 2>&1 || { tail -30 gpurun_out/r7f/gpu_suite.txt; exit 1; }
tail -3 gpurun_out/r7f/gpu_suite.txt
SY11_WGRAD_STREAM=0 bash tools/prof_bench.sh stats r4ser > /dev/null
python tools/trace_summary.py gpurun_out/prof_r4ser/run_kernel_trace.csv gpurun_out/r7f/z_serial 5 | tail -12
cp gpurun_out/prof_r4ser/bench.json gpurun_out/r7f/z_serial_bench.json
bash tools/prof_bench.sh stats r4fin > /dev/null
python tools/trace_summary.py gpurun_out/prof_r4fin/run_kernel_trace.csv gpurun_out/r7f/z_final 5 | tail -12
cp gpurun_out/prof_r4fin/bench.json gpurun_out/r7f/z_final_bench.json
rm -f gpurun_out/prof_r4ser/run_kernel_trace.csv gpurun_out/prof_r4fin/run_kernel_trace.csv
echo stats done
bash tools/prof_bench.sh pmc r4pmc > /dev/null
python tools/pmc_summary.py gpurun_out/prof_r4pmc gpurun_out/r7f 2 | tail -12
rm -rf gpurun_out/prof_r4pmc/FETCH_SIZE gpurun_out/prof_r4pmc/WRITE_SIZE
bash tools/prof_bench.sh predict r4pred > /dev/null
cp gpurun_out/prof_r4pred/bench.json gpurun_out/r7f/predict_val_bench.json
f=$(ls gpurun_out/prof_r4pred/*kernel_stats.csv | head -1); cp "$f" gpurun_out/r7f/predict_val_kernel_stats.csv
rm -f gpurun_out/prof_r4pred/*kernel_trace.csv
echo all done
timeout -k 10 200 python tools/soak.py > gpurun_out/r7f/soak_300_steps.txt 2>&1 || true
tail -3 gpurun_out/r7f/soak_300_steps.txt
SY11_DETERMINISTIC=1 SY11_WGRAD_STREAM=0 bash tools/prof_bench.sh stats r4det > /dev/null
python tools/trace_summary.py gpurun_out/prof_r4det/run_kernel_trace.csv gpurun_out/r7f/z_ordered 5 | tail -8
rm -f gpurun_out/prof_r4det/run_kernel_trace.csv
