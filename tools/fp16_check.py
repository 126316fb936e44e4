import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "spectrogram-yolov11_amd"))
from types import SimpleNamespace
import torch
from oracle import loss_ref, yolo11_ref as R
from sy11.nn.tasks import DetectionModel
sd = R.seeded_state_dict(R.empty_state_dict(R.resolve_graph("n", nc=80)), seed=1)
torch.manual_seed(3)
for nb in (2, 8):
    img = torch.rand(nb, 3, 128, 128)
    batch = {"img": img.cuda(), "batch_idx": torch.tensor([0., 0., float(nb - 1)]).cuda(), "cls": torch.tensor([[3.], [17.], [60.]]).cuda(),
             "bboxes": torch.tensor([[0.4, 0.4, 0.5, 0.4], [0.6, 0.65, 0.3, 0.5], [0.5, 0.5, 0.7, 0.6]]).cuda()}
    osd = {k: v.clone() for k, v in sd.items()}
    maps = R.forward(osd, R.resolve_graph("n", nc=80), img, train=True)
    ol, oi = loss_ref.detection_loss(maps, {k: v.cpu() for k, v in batch.items()}, nc=80)
    print("nb", nb, "oracle", ol.item(), oi.tolist())
    for dt in (torch.float32, torch.float16, torch.float16):
        m = DetectionModel("yolo11n.yaml", nc=80, verbose=False)
        m.args = SimpleNamespace(box=7.5, cls=0.5, dfl=1.5)
        m.load_state_dict(sd); m._sy11_dtype = dt
        m = m.cuda().train()
        l, it = m(batch)
        print("   ", dt, l.item(), it.tolist())
