"""List every sy11_copy2d launch of one yolo11s training step (64 x 3 x 640 x 640, f16): shape, strides, accumulate flag, caller."""
import sys, traceback
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT / "spectrogram-yolov11_amd"))
sys.path.insert(0, str(ROOT))
import torch
from sy11 import ops
from sy11.engine.trainer import DetectionTrainer
from sy11.nn.tasks import DetectionModel
import bench

dev = torch.device("cuda", 0)
model = DetectionModel("yolo11s.yaml", nc=80, verbose=False)
tr = DetectionTrainer(model, batch_size=64, device=dev, overrides={"amp": True}, world_size=1, producer=None, graphs=False)
data = {"img": torch.rand(64, 3, 640, 640, device=dev)}
labels = bench.synthetic_labels(64, 100, dev, nc=80)
tr.train_step({**data, **labels})
log = []
orig = ops.copy2d
def spy(src, dst, accumulate=False):
    fr = [f for f in traceback.extract_stack()[:-1] if "sy11" in f.filename][-1]
    log.append((tuple(src.shape), src.dtype, ops.view_ld(src), ops.view_ld(dst), bool(accumulate), f"{Path(fr.filename).name}:{fr.lineno}"))
    return orig(src, dst, accumulate=accumulate)
ops.copy2d = spy
import sy11.nn.modules.conv as C, sy11.nn.modules.block as Bk
for mod in (C, Bk):
    if hasattr(mod, "ops"):
        mod.ops.copy2d = spy
tr.train_step({**data, **labels})
torch.cuda.synchronize()
for e in sorted(log, key=lambda e: -e[0][0] * e[0][1] * e[0][2] * e[0][3]):
    n = e[0][0] * e[0][1] * e[0][2] * e[0][3]
    print(f"{str(e[0]):24s} {str(e[1]):14s} ld {e[2]:4d} -> {e[3]:4d} acc {int(e[4])}  {n * 2 * (3 if e[4] else 2) / 1e6:7.1f} MB  {e[5]}")
