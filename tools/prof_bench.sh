# rocprofv3 profiles of the default bench command; the caller copies the summaries into profiles/.
#   bash tools/prof_bench.sh stats <tag>   kernel trace + per-kernel statistics (graph replay, the bench as the driver runs it)
#   bash tools/prof_bench.sh predict <tag> kernel statistics of the predict / val leg alone (fused eval forward + Detect decode + batched NMS, bs 64)
#   bash tools/prof_bench.sh pmc <tag>     HBM traffic counters, one pass per counter (eager launches: counters are per dispatch)
set -e
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"     # repo root: gpurun exports it; else derived from this script's path
MODE=$1; TAG=$2
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd $GRAFT_REPO_ROOT
if [ "$MODE" = predict ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o run -- python3 bench.py --leg predict_val --steps 10 --warmup 3 > $OUT/bench.json 2> $OUT/bench.err
elif [ "$MODE" = stats ]; then
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o run -- python3 bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-roofline --no-fwd-leg --no-extras > $OUT/bench.json 2> $OUT/bench.err
else
  # the counter passes must run the kernels a normal run picks: record the tuner's picks first, replay them with measuring off
  SY11_TUNE_SAVE=$OUT/picks.bin python3 bench.py --steps 2 --warmup 3 --no-cpu-baseline --no-roofline --no-fwd-leg --no-extras > $OUT/bench_plain.json 2> $OUT/bench_plain.err
  export SY11_TUNE_LOAD=$OUT/picks.bin SY11_TUNE=0
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -o run -- python3 bench.py --steps 2 --warmup 3 --no-cpu-baseline --no-roofline --no-fwd-leg --no-extras --no-graphs > $OUT/bench_$c.json 2> $OUT/bench_$c.err
  done
fi
find $OUT -type f | head -30
