"""Time the C2PSA attention core at the bench shape (yolo11s 640^2, batch 64: 400 tokens, 4 heads).   python tools/attention_micro.py"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "spectrogram-yolov11_amd"))
sys.path.insert(0, str(ROOT / "tools"))
import torch
from sy11 import ops
from bn_sweep import timed

B, H, W, heads, kd, hd = 64, 20, 20, 4, 32, 64
N = H * W
dt = torch.float16
qkv = torch.randn(B, H, W, heads * (2 * kd + hd), device="cuda", dtype=dt)
o = torch.empty(B, H, W, heads * hd, device="cuda", dtype=dt)
p = torch.empty(B, heads, N, N, device="cuda", dtype=torch.float32)
do = torch.randn_like(o)
dq = torch.empty_like(qkv)
ws = torch.empty(B * heads * N * N, device="cuda", dtype=torch.float32)
ops.attention_fwd(qkv, heads, kd, hd, o, p)
print(f"forward                       {timed(lambda: ops.attention_fwd(qkv, heads, kd, hd, o, p), 20, False) * 1e3:7.1f} us")
print(f"backward (P read twice)       {timed(lambda: ops.attention_bwd(qkv, heads, kd, hd, p, do, dq, ws), 20, False) * 1e3:7.1f} us")
print(f"backward (row sums from o)    {timed(lambda: ops.attention_bwd(qkv, heads, kd, hd, p, do, dq, ws, o=o), 20, False) * 1e3:7.1f} us")
a = dq.clone()
ops.attention_bwd(qkv, heads, kd, hd, p, do, dq, ws)
print("max |difference| between the two forms:", float((a.float() - dq.float()).abs().max()), "of", float(dq.float().abs().max()))
