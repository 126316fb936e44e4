"""Aggregate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/prof_bench.sh pmc) into per-kernel HBM traffic.

Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE (KiB) counts 128-byte requests at
64 bytes -> doubled; WRITE_SIZE (KiB) is exact for 16-byte stores and float atomics.  Output: a CSV per kernel and
profiles/<round>/traffic.json with bytes per launch for the C-ABI families bench.py reports.
    python tools/pmc_summary.py gpurun_out/prof_p1 profiles/r01 <steps profiled>"""
import csv, json, re, sys
from collections import defaultdict
from pathlib import Path

src, dst, steps = Path(sys.argv[1]), Path(sys.argv[2]), int(sys.argv[3])
acc = defaultdict(lambda: {"n": 0, "FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0})
for counter in ("FETCH_SIZE", "WRITE_SIZE"):
    with open(src / counter / "run_counter_collection.csv") as f:
        allrows = [r for r in csv.DictReader(f) if r["Counter_Name"] == counter]
        allrows.sort(key=lambda r: int(r["Dispatch_Id"]))
        # keep only the last `steps` training steps (a step starts with the STFT kernel): the first eager step also contains the
        # tile autotuner's measurement launches
        marks = [i for i, r in enumerate(allrows) if "stft_logmel" in r["Kernel_Name"]]
        start = marks[-steps] if len(marks) >= steps else 0
        seen = defaultdict(int)
        for row in allrows[start:]:
            k = row["Kernel_Name"]
            acc[k][counter] += float(row["Counter_Value"])
            seen[k] += 1
        for k, n in seen.items():
            acc[k]["n"] = max(acc[k]["n"], n)


def family(k):
    if "wgrad16_kernel" in k or re.search(r"\bwgrad_kernel", k):
        return "sy11_conv2d_wgrad"
    if "igemm" in k:
        return "sy11_conv2d_fwd+dgrad"
    for f in ("bn_bwd_reduce", "bn_bwd_apply", "bn_act_fwd", "stem_fwd", "stem_wgrad", "dw3x3", "dwconv_wgrad", "stft_logmel", "attention"):
        if f in k:
            return f
    return None


dst.mkdir(parents=True, exist_ok=True)
rows, fam = [], defaultdict(lambda: {"n": 0, "bytes": 0.0})
for k, v in acc.items():
    b = v["FETCH_SIZE"] * 1024 * 2 + v["WRITE_SIZE"] * 1024
    rows.append((b, k, v))
    f = family(k)
    if f:
        fam[f]["n"] += v["n"]
        fam[f]["bytes"] += b
rows.sort(reverse=True)
with open(dst / "f_pmc_hbm_traffic_by_kernel.csv", "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "launches", "FETCH_SIZE_KiB_sum", "WRITE_SIZE_KiB_sum", "corrected_bytes_per_launch", "corrected_GB_per_step"])
    for b, k, v in rows[:80]:
        w.writerow([k[:160], v["n"], round(v["FETCH_SIZE"], 1), round(v["WRITE_SIZE"], 1), round(b / max(v["n"], 1)), round(b / steps / 1e9, 3)])
out = {"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --steps 2 --warmup 3 --no-graphs; FETCH x2 (gfx950), WRITE x1",
       "steps_profiled": steps,
       "families": {k: {"launches": v["n"], "bytes_per_launch": v["bytes"] / max(v["n"], 1), "GB_per_step": v["bytes"] / steps / 1e9}
                    for k, v in sorted(fam.items())}}
# bench.py looks the three conv entry points up by name: fwd and dgrad share the igemm kernel
out["families"]["sy11_conv2d_fwd"] = out["families"]["sy11_conv2d_dgrad"] = out["families"].get("sy11_conv2d_fwd+dgrad", {})
(dst / "traffic.json").write_text(json.dumps(out, indent=1))
for k, v in sorted(out["families"].items()):
    if v:
        print(f"{k:28s} {v['launches']:5d} launches  {v['bytes_per_launch'] / 1e6:9.1f} MB/launch  {v['GB_per_step']:7.2f} GB/step")
