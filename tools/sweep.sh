for cfg in "64 40 40 384 256 1 1" "64 80 80 128 128 1 1" "64 40 40 64 64 3 1" "64 20 20 128 128 3 1" "64 80 80 256 256 3 2" "64 160 160 32 16 3 1"; do
for d in 0 1 2 3 4; do echo -n "debug=$d "; SY11_TUNE=0 SY11_IGEMM_DEBUG=$d python tools/conv_micro.py $cfg fwd 50 2>/dev/null; done; done
