# memory-side counters of one conv problem (tools/conv_micro.py): L1/L2 latency, stalls, hit rates
#   MODE=fwd bash tools/pmc_mem.sh <tag> B H W C N k s
set -e
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"     # repo root: gpurun exports it; else derived from this script's path
TAG=$1; shift
ARGS="$1 $2 $3 $4 $5 $6 $7"
export SY11_TUNE=0
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcm_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd $GRAFT_REPO_ROOT
i=0
for set in "TCP_TCC_READ_REQ TCP_TCC_READ_REQ_LATENCY TCP_TCP_LATENCY TCP_TOTAL_ACCESSES" "TCP_PENDING_STALL_CYCLES TCP_READ_TAGCONFLICT_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TCP_TOTAL_CACHE_ACCESSES" "TCC_HIT TCC_MISS TCC_REQ TCC_TAG_STALL" "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LEVEL_WAVES SQ_IFETCH" "TCP_UTCL1_TRANSLATION_MISS TCP_UTCL1_TRANSLATION_HIT TCP_UTCL1_REQUEST TCP_CACHE_MISS"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o run -- python3 tools/conv_micro.py $ARGS ${MODE:-fwd} 5 > $OUT/p$i.log 2>&1 || echo "pass $i failed: $set"
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob("$OUT/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if not any(s in r["Kernel_Name"] for s in ("igemm", "wgrad", "halo3x3")): continue
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for c in sorted(acc):
    print(f"   {c:36s} {acc[c] / n[c]:18.0f}  per launch")
PY
