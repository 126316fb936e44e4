import sys, torch
sys.path.insert(0, "spectrogram-yolov11_amd")
from types import SimpleNamespace
from sy11.nn.tasks import DetectionModel
for sc, n in (("n", 2624080), ("s", 9458752), ("m", 20114688), ("l", 25372160), ("x", 56966176)):
    m = DetectionModel(f"yolo11{sc}.yaml", nc=80, verbose=False)
    assert sum(p.numel() for p in m.parameters()) == n, (sc, sum(p.numel() for p in m.parameters()))
    m.args = SimpleNamespace(box=7.5, cls=0.5, dfl=1.5)
    m = m.cuda().train()
    for dt in (torch.float32, torch.float16):
        m._sy11_dtype = dt
        b = {"img": torch.rand(2, 3, 64, 64).cuda(), "batch_idx": torch.tensor([0., 1.]).cuda(), "cls": torch.tensor([[1.], [2.]]).cuda(),
             "bboxes": torch.tensor([[0.5, 0.5, 0.4, 0.4], [0.4, 0.6, 0.3, 0.3]]).cuda()}
        try:
            loss, items = m(b)
            loss.backward()
            print(sc, dt, "ok", float(loss))
        except Exception as e:
            print(sc, dt, "FAIL", str(e)[:200])
