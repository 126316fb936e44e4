"""Robustness probe: yolo11s train + eval(fused) + NMS at odd batch sizes / resolutions / dtypes."""
import sys, torch
sys.path.insert(0, "spectrogram-yolov11_amd")
from types import SimpleNamespace
from sy11.nn.tasks import DetectionModel
from sy11.utils.ops import non_max_suppression
for (B, H, W, dt) in ((1, 416, 416, torch.float16), (3, 320, 320, torch.float32), (5, 480, 640, torch.float16), (2, 64, 96, torch.bfloat16), (7, 32, 32, torch.float16),
                      (2, 1280, 1280, torch.float16)):
    m = DetectionModel("yolo11s.yaml", nc=3, verbose=False)
    m.args = SimpleNamespace(box=7.5, cls=0.5, dfl=1.5)
    m = m.cuda().train()
    m._sy11_dtype = dt
    b = {"img": torch.rand(B, 3, H, W).cuda(), "batch_idx": torch.tensor([0., float(B - 1)]).cuda(), "cls": torch.tensor([[1.], [2.]]).cuda(),
         "bboxes": torch.tensor([[0.5, 0.5, 0.4, 0.4], [0.4, 0.6, 0.3, 0.3]]).cuda()}
    try:
        loss, items = m(b)
        loss.backward()
        ok = all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)
        m.eval()
        with torch.no_grad():
            y, _ = m(b["img"])
            m.fuse()
            yf, _ = m(b["img"])
            det = non_max_suppression(yf, 0.001, 0.7, max_det=50)
        print(B, H, W, dt, "ok" if ok else "NONFINITE", round(float(loss.detach()), 4), tuple(y.shape), float((y - yf).abs().max()), [len(d) for d in det])
    except Exception as e:
        print(B, H, W, dt, "FAIL", str(e)[:300])
