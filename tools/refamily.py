"""Re-derive `<name>_per_step_families.csv` from `<name>_per_step_kernels.csv` with bench.FAMILIES as it is NOW (a kernel symbol added to a
family after the trace was summarised; the raw trace is not kept).  The span / in-flight rows are carried over.
    python tools/refamily.py profiles/r04/z_serial"""
import csv, sys
from collections import defaultdict
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from bench import FAMILIES  # noqa: E402


def family(k):
    for fam, spec in FAMILIES.items():
        if any(s in k for s in spec["symbols"]):
            return fam
    return "other"


base = sys.argv[1]
rows = list(csv.DictReader(open(base + "_per_step_kernels.csv")))
for r in rows:
    r["family"] = family(r["kernel"])
with open(base + "_per_step_kernels.csv", "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
    w.writeheader()
    w.writerows(rows)
old = list(csv.reader(open(base + "_per_step_families.csv")))
tail = [r for r in old if r and r[0].startswith("(")]
fam = defaultdict(lambda: [0.0, 0.0])
for r in rows:
    fam[r["family"]][0] += float(r["launches_per_step"])
    fam[r["family"]][1] += float(r["ms_per_step"])
busy = sum(v[1] for v in fam.values())
with open(base + "_per_step_families.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(old[0])
    for k, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
        w.writerow([k, round(c, 2), round(t, 4), round(t / c * 1e3, 2), round(t / busy, 4)])
    w.writerows(tail)
for k, (c, t) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:28s} {c:7.1f} launches  {t:8.3f} ms/step")
