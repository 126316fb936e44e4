"""Ablations of the 8-wave pipeline (igemm cfg 20) against the 4-wave 256x128 tile (cfg 3): SY11_IGEMM_DEBUG is read once per
process, so every mode is a child process.   python tools/igemm8_ablate.py [cfg ...]"""
import os, subprocess, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
cfgs = [a for a in sys.argv[1:]] or ["3", "20"]
MODES = {0: "full", 1: "no DMA after the prologue", 2: "no MFMAs", 6: "cfg 20: no pixel-row pieces", 7: "cfg 20: no filter-row pieces", 8: "cfg 20: copies and barriers only"}
sel = [int(a[1:]) for a in sys.argv[1:] if a.startswith("d")] or [0, 1, 2]
cfgs = [a for a in cfgs if not a.startswith("d")] or ["3", "20"]
for dbg, what in ((d, MODES[d]) for d in sel):
    print(f"--- SY11_IGEMM_DEBUG={dbg} ({what})", flush=True)
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "igemm_probe.py"), *cfgs], env=dict(os.environ, SY11_IGEMM_DEBUG=str(dbg)),
                       capture_output=True, text=True)
    print("\n".join(l for l in r.stdout.splitlines() if "->" in l), flush=True)
    if r.returncode:
        print(r.stderr[-2000:])
