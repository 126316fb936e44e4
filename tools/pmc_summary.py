"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/prof_bench.sh pmc) into per-kernel and per-family HBM traffic
PER TRAINING STEP.

Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE (KiB) counts 128-byte requests at
64 bytes -> doubled; WRITE_SIZE (KiB) is exact for 16-byte stores and float atomics.  Only the last <steps> training steps of
each pass are used (a step starts with the STFT kernel), so the tile autotuner's measurement launches of the first eager step
never enter.  Families and their kernel symbols are the ones bench.py reports (bench.FAMILIES).
    python tools/pmc_summary.py gpurun_out/prof_<tag> profiles/r02 <steps profiled>"""
import csv, json, sys
from collections import defaultdict
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from bench import FAMILIES  # noqa: E402

src, dst, steps = Path(sys.argv[1]), Path(sys.argv[2]), int(sys.argv[3])
acc = defaultdict(lambda: {"n": 0, "FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0})
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    with open(src / counter / "run_counter_collection.csv") as f:
        allrows = [r for r in csv.DictReader(f) if r["Counter_Name"] == counter]
        allrows.sort(key=lambda r: int(r["Dispatch_Id"]))
        marks = [i for i, r in enumerate(allrows) if "stft_logmel" in r["Kernel_Name"]]
        start = marks[-steps] if len(marks) >= steps else 0
        seen = defaultdict(int)
        for row in allrows[start:]:
            k = row["Kernel_Name"]
            acc[k][counter] += float(row["Counter_Value"])
            seen[k] += 1
        if counter == "FETCH_SIZE":                 # launch counts from ONE pass (both passes replay the same recorded tile picks)
            for k, n in seen.items():
                acc[k]["n"] = n


def family(k):
    for fam, spec in FAMILIES.items():
        if any(s in k for s in spec["symbols"]):
            return fam
    return None


dst.mkdir(parents=True, exist_ok=True)
rows, fam = [], defaultdict(lambda: {"n": 0, "bytes": 0.0, "fetch": 0.0, "write": 0.0})
total = 0.0
for k, v in acc.items():
    fb, wb = v["FETCH_SIZE"] * 1024 * 2, v["WRITE_SIZE"] * 1024
    b = fb + wb
    total += b
    rows.append((b, k, v))
    f = family(k)
    if f:
        fam[f]["n"] += v["n"]
        fam[f]["bytes"] += b
        fam[f]["fetch"] += fb
        fam[f]["write"] += wb
rows.sort(reverse=True)
with open(dst / "pmc_hbm_traffic_by_kernel.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "launches_per_step", "FETCH_SIZE_KiB_sum", "WRITE_SIZE_KiB_sum", "corrected_bytes_per_launch", "corrected_GB_per_step"])
    for b, k, v in rows[:100]:
        w.writerow([k[:160], round(v["n"] / steps, 2), round(v["FETCH_SIZE"], 1), round(v["WRITE_SIZE"], 1), round(b / max(v["n"], 1)), round(b / steps / 1e9, 3)])
import subprocess  # noqa: E402


def _git(*args):
    try:
        return subprocess.run(["git", *args], capture_output=True, text=True, cwd=Path(__file__).resolve().parents[1]).stdout.strip()
    except Exception:
        return ""


sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import bench as _bench
out = {"commit": _git("rev-parse", "HEAD"), "csrc_sha16": _bench.kernel_source_sha16(), "worktree_dirty": bool(_git("status", "--porcelain", "--", "spectrogram-yolov11_amd", "bench.py")),
       "commands": ["SY11_TUNE_SAVE=$OUT/picks.bin python3 bench.py --steps 2 --warmup 3 --no-cpu-baseline --no-roofline --no-fwd-leg --no-extras",
                    "SY11_TUNE_LOAD=$OUT/picks.bin SY11_TUNE=0 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/FETCH_SIZE -o run -- python3 bench.py --steps 2 --warmup 3 --no-cpu-baseline --no-roofline --no-fwd-leg --no-extras --no-graphs",
                    "SY11_TUNE_LOAD=$OUT/picks.bin SY11_TUNE=0 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/WRITE_SIZE -o run -- python3 bench.py --steps 2 --warmup 3 --no-cpu-baseline --no-roofline --no-fwd-leg --no-extras --no-graphs",
                    f"python tools/pmc_summary.py {src} {dst} {steps}"],
       "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes): python3 bench.py --steps 2 --warmup 3 --no-graphs "
                 "--no-cpu-baseline --no-roofline --no-fwd-leg --no-extras; last steps only; FETCH x2 (gfx950), WRITE x1",
       "steps_profiled": steps, "total_GB_per_step": round(total / steps / 1e9, 3),
       "families": {k: {"launches_per_step": round(v["n"] / steps, 2), "GB_per_step": round(v["bytes"] / steps / 1e9, 4),
                        "fetch_GB_per_step": round(v["fetch"] / steps / 1e9, 4), "write_GB_per_step": round(v["write"] / steps / 1e9, 4),
                        "kernel_symbols": list(FAMILIES[k]["symbols"])}
                    for k, v in sorted(fam.items())}}
(dst / "traffic.json").write_text(json.dumps(out, indent=1))
print(f"total {out['total_GB_per_step']} GB/step")
for k, v in sorted(out["families"].items()):
    print(f"{k:28s} {v['launches_per_step']:7.1f} launches/step  {v['GB_per_step']:7.2f} GB/step")
