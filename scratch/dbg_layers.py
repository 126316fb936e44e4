import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'spectrogram-yolov11_amd')
import torch, torch.nn.functional as F
from tests.test_model_gpu import tiny_model
from oracle import yolo11_ref as R
m = tiny_model().train()
layers = R.resolve_graph("t", nc=4)
sd = R.closed_form_state_dict(R.empty_state_dict(layers))
img = R.closed_form("in.model_t", (2, 3, 64, 64), "input")
# oracle per-layer
saved = []; x = img
ins = []
for L in layers:
    i, f, kind = L["i"], L["f"], L["kind"]
    p = f"model.{i}."
    if f != -1:
        x = saved[f] if isinstance(f, int) else [x if j == -1 else saved[j] for j in f]
    ins.append(x)
    if kind == "Conv": x = R.conv_bn_act(sd, p, x, L["k"], L["s"], train=True)
    elif kind == "C3k2": x = R.c3k2(sd, p, x, L["c2"], L["n"], L["c3k"], L["e"], True, True)
    elif kind == "SPPF": x = R.sppf(sd, p, x, L["k"], True)
    elif kind == "C2PSA": x = R.c2psa(sd, p, x, L["n"], L["e"], True)
    elif kind == "Upsample": x = F.interpolate(x, scale_factor=2.0, mode="nearest")
    elif kind == "Concat": x = torch.cat(x, 1)
    elif kind == "Detect": x = R.detect_head(sd, p, x, L["nc"], True)
    saved.append(x)
for L, xin, xout in zip(layers, ins, saved):
    mod = m.model[L["i"]]
    if L["kind"] == "Upsample":
        continue
    xi = [t.cuda() for t in xin] if isinstance(xin, list) else xin.cuda()
    with torch.no_grad():
        y = mod(xi)
    ys = y if isinstance(y, (list, tuple)) else [y]
    xo = xout if isinstance(xout, list) else [xout]
    for a, b in zip(ys, xo):
        err = (a.float().cpu() - b).abs().max().item()
        print(L["i"], L["kind"], tuple(b.shape), "err %.3e scale %.3e" % (err, b.abs().max().item()))
