# SQ counters of one wgrad problem (tools/conv_micro.py ... wgrad): where do the wave cycles go?
#   bash tools/pmc_wgrad.sh <tag> B H W C N k s [SY11_WGRAD_CFG]
set -e
: "${GRAFT_REPO_ROOT:=$(cd "$(dirname "$0")/.." && pwd)}"     # repo root: gpurun exports it; else derived from this script's path
TAG=$1; shift
ARGS="$1 $2 $3 $4 $5 $6 $7"
export SY11_TUNE=0
[ -n "$8" ] && export SY11_WGRAD_CFG=$8
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES --output-format csv -d $OUT/a -o run -- python3 tools/conv_micro.py $ARGS ${MODE:-wgrad} 5 > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_ANY --output-format csv -d $OUT/b -o run -- python3 tools/conv_micro.py $ARGS ${MODE:-wgrad} 5 > $OUT/b.log 2>&1
python3 - <<PY
import csv, glob, collections
for part in ("a", "b"):
    f = glob.glob("$OUT/%s/**/*counter_collection.csv" % part, recursive=True)
    if not f:
        print(part, "no counter file"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f[0])):
        if "wgrad" not in r["Kernel_Name"] and "igemm" not in r["Kernel_Name"]: continue
        acc[r["Kernel_Name"][:40]][r["Counter_Name"]] += float(r["Counter_Value"]); n[(r["Kernel_Name"][:40], r["Counter_Name"])] += 1
    for k, d in acc.items():
        print(part, k)
        for c, v in sorted(d.items()):
            print(f"   {c:28s} {v / n[(k, c)]:16.0f}  per launch")
PY
