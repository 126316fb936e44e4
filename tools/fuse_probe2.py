import sys, torch
sys.path.insert(0, "spectrogram-yolov11_amd"); sys.path.insert(0, ".")
from types import SimpleNamespace
from sy11.nn.tasks import DetectionModel
from oracle import yolo11_ref as R
B, H, W = 3, 320, 320
m = DetectionModel("yolo11s.yaml", nc=3, verbose=False)
m.args = SimpleNamespace(box=7.5, cls=0.5, dfl=1.5)
m = m.cuda().train()
img = torch.rand(B, 3, H, W)
b = {"img": img.cuda(), "batch_idx": torch.tensor([0., float(B - 1)]).cuda(), "cls": torch.tensor([[1.], [2.]]).cuda(), "bboxes": torch.tensor([[0.5, 0.5, 0.4, 0.4], [0.4, 0.6, 0.3, 0.3]]).cuda()}
loss, items = m(b); loss.backward()
m.eval()
sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
layers = R.resolve_graph("s", nc=3)
with torch.no_grad():
    y, maps = m(b["img"])
    oy, omaps = R.forward(sd, layers, img, train=False)
    print("unfused vs oracle", float((y.cpu() - oy).abs().max()), [float((a.cpu() - o).abs().max()) for a, o in zip(maps, omaps)])
    m.fuse()
    yf, mapsf = m(b["img"])
    oyf, omapsf = R.forward(R.fuse_state_dict(sd), layers, img, train=False, fused=True)
    print("fused vs oracle-fused", float((yf.cpu() - oyf).abs().max()), [float((a.cpu() - o).abs().max()) for a, o in zip(mapsf, omapsf)])
    print("oracle fused vs unfused", float((oyf - oy).abs().max()))
    d = (yf.cpu() - oyf).abs()
    print("rows max err", d.amax((0, 2)))
