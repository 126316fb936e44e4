set -e
SY11_IGEMM_BM256_WG=1 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv or igemm or dgrad" 2>&1 | tail -3
run() { echo "$@"; env "$@" python tools/conv_sweep.py fwd,dgrad -v 2>/dev/null > gpurun_out/sw_$N.txt; tail -1 gpurun_out/sw_$N.txt; N=$((N+1)); }
N=0
run SY11_IGEMM_BM256_WG=100000000
run SY11_IGEMM_BM256_WG=512
run SY11_IGEMM_BM256_WG=256
run SY11_IGEMM_BM256_WG=128
