"""rocprofv3 --kernel-trace of the default bench command -> per-STEP kernel statistics of the graph-replayed steps only.

The raw `*_kernel_stats.csv` of a bench run mixes three things: the first eager step with the tile autotuner's measurement launches,
the second eager step, and the replayed steps.  This script cuts the trace at the STFT kernel (first kernel of a step), keeps the
last <n> complete steps, and writes (a) one row per kernel symbol: launches per step, total and average duration per step, and
(b) one row per kernel family of bench.FAMILIES — the numbers `roofline.family_ms_per_step` of the bench line is to be compared with.
    python tools/trace_summary.py gpurun_out/prof_<tag>/run_kernel_trace.csv profiles/r02/<name> [steps=5]"""
import csv, re, sys
from collections import defaultdict
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from bench import FAMILIES  # noqa: E402

src, dst = Path(sys.argv[1]), sys.argv[2]
nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
rows = list(csv.DictReader(open(src)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "stft_logmel" in r["Kernel_Name"]]
# the train steps of the timed region are followed by the forward-only legs (no stft) — keep steps whose kernel count equals the mode
sizes = [marks[i + 1] - marks[i] for i in range(len(marks) - 1)]
mode = max(set(sizes), key=sizes.count)
good = [i for i in range(len(marks) - 1) if sizes[i] == mode][-nsteps:]
sym = defaultdict(lambda: [0, 0])
span = busy = 0
union = two = 0                       # time with >= 1 / >= 2 kernels in flight (the engine's second stream: filter gradients)
queues = set()
for i in good:
    step = rows[marks[i]:marks[i + 1]]
    span += int(step[-1]["End_Timestamp"]) - int(step[0]["Start_Timestamp"])
    ev = sorted([(int(r["Start_Timestamp"]), 1) for r in step] + [(int(r["End_Timestamp"]), -1) for r in step])
    live, last = 0, None
    for t, d in ev:
        if last is not None:
            union += (t - last) if live >= 1 else 0
            two += (t - last) if live >= 2 else 0
        live += d
        last = t
    queues |= {r.get("Queue_Id") for r in step}
    for r in step:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        busy += d
        s = sym[r["Kernel_Name"]]
        s[0] += 1
        s[1] += d
n = len(good)


def short(k):
    return re.sub(r"\s+", " ", k)[:150]


def family(k):
    for fam, spec in FAMILIES.items():
        if any(s in k for s in spec["symbols"]):
            return fam
    return "other"


with open(dst + "_per_step_kernels.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "family", "launches_per_step", "ms_per_step", "avg_us_per_launch"])
    for k, (c, t) in sorted(sym.items(), key=lambda kv: -kv[1][1]):
        w.writerow([short(k), family(k), round(c / n, 2), round(t / n / 1e6, 4), round(t / c / 1e3, 2)])
fam = defaultdict(lambda: [0, 0])
for k, (c, t) in sym.items():
    fam[family(k)][0] += c
    fam[family(k)][1] += t
with open(dst + "_per_step_families.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["family", "kernel_launches_per_step", "ms_per_step", "avg_us_per_launch", "share_of_step"])
    for k, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        w.writerow([k, round(c / n, 2), round(t / n / 1e6, 4), round(t / c / 1e3, 2), round(t / busy, 4)])
    w.writerow(["(step span)", round(mode, 0), round(span / n / 1e6, 4), "", round(busy / span, 4)])
    w.writerow(["(>= 1 kernel in flight)", "", round(union / n / 1e6, 4), "", round(union / span, 4)])
    w.writerow(["(>= 2 kernels in flight)", len(queues), round(two / n / 1e6, 4), "", round(two / span, 4)])
print(f"{n} replayed steps of {mode} kernels: span {span / n / 1e6:.3f} ms, sum of kernel durations {busy / n / 1e6:.3f} ms, "
      f">= 1 kernel in flight {union / n / 1e6:.3f} ms, >= 2 in flight {two / n / 1e6:.3f} ms, {len(queues)} queue(s)")
for k, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:28s} {c / n:7.1f} launches  {t / n / 1e6:8.3f} ms/step")
