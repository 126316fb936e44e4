for cfg in "64 160 160 16 32 3 1" "64 160 160 32 16 3 1" "64 80 80 32 64 3 1" "64 320 320 32 64 3 2" "64 160 160 128 128 3 2"; do
for m in fwd fwd_nostats; do python tools/conv_micro.py $cfg $m 30 2>/dev/null | tail -1; done; done
