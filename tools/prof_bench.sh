# rocprofv3 kernel trace + stats of the default bench (graph replay), summaries copied by the caller into profiles/
set -e
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$1
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d $OUT -o run -- python3 bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-roofline > $OUT/bench.json 2> $OUT/bench.err
ls $OUT $OUT/* | head -30
