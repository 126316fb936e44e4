import sys, torch
sys.path.insert(0, "spectrogram-yolov11_amd"); sys.path.insert(0, ".")
from types import SimpleNamespace
from sy11.nn.tasks import DetectionModel
from oracle import yolo11_ref as R
torch.manual_seed(0)
m = DetectionModel("yolo11n.yaml", nc=3, verbose=False)
m.args = SimpleNamespace(box=7.5, cls=0.5, dfl=1.5)
m = m.cuda().train()
img = torch.rand(2, 3, 64, 64)
b = {"img": img.cuda(), "batch_idx": torch.tensor([0., 1.]).cuda(), "cls": torch.tensor([[1.], [2.]]).cuda(), "bboxes": torch.tensor([[0.5, 0.5, 0.4, 0.4], [0.4, 0.6, 0.3, 0.3]]).cuda()}
for step in range(2):
    if step == 1:
        loss, items = m(b); loss.backward()
    m.eval()
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    layers = R.resolve_graph("n", nc=3)
    with torch.no_grad():
        y, _ = m(b["img"])
        oy, _ = R.forward(sd, layers, img, train=False)
        print("step", step, "unfused vs oracle", float((y.cpu() - oy).abs().max()), "ymax", float(oy.abs().max()))
        import copy
        mf = copy.deepcopy(m); mf.fuse()
        yf, _ = mf(b["img"])
        oyf, _ = R.forward(R.fuse_state_dict(sd), layers, img, train=False, fused=True)
        print("        fused vs oracle-fused", float((yf.cpu() - oyf).abs().max()), " oracle fused vs unfused", float((oyf - oy).abs().max()))
    m.train()

print("---- in-place fuse after an eval forward (the shapes_probe flow), yolo11s")
m = DetectionModel("yolo11s.yaml", nc=3, verbose=False)
m.args = SimpleNamespace(box=7.5, cls=0.5, dfl=1.5)
m = m.cuda().train()
loss, items = m(b); loss.backward()
m.eval()
sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
layers = R.resolve_graph("s", nc=3)
with torch.no_grad():
    y, _ = m(b["img"])
    oy, _ = R.forward(sd, layers, img, train=False)
    print("unfused vs oracle", float((y.cpu() - oy).abs().max()))
    m.fuse()
    yf, _ = m(b["img"])
    print("fused vs oracle", float((yf.cpu() - oy).abs().max()), "fused vs unfused", float((yf - y).abs().max()))
    idx = (yf - y).abs().flatten().argmax().item()
    print("worst element", idx, float(y.flatten()[idx]), float(yf.flatten()[idx]), float(oy.flatten()[idx]))
