import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'spectrogram-yolov11_amd')
import torch, torch.nn.functional as F
from tests.test_model_gpu import tiny_model
from oracle import yolo11_ref as R
layers = R.resolve_graph("t", nc=4)
img = R.closed_form("in.model_t", (2, 3, 64, 64), "input")
def oracle_layers():
    sd = R.closed_form_state_dict(R.empty_state_dict(layers))
    saved = []; x = img
    for L in layers:
        i, f, kind = L["i"], L["f"], L["kind"]
        p = f"model.{i}."
        if f != -1:
            x = saved[f] if isinstance(f, int) else [x if j == -1 else saved[j] for j in f]
        if kind == "Conv": x = R.conv_bn_act(sd, p, x, L["k"], L["s"], train=True)
        elif kind == "C3k2": x = R.c3k2(sd, p, x, L["c2"], L["n"], L["c3k"], L["e"], True, True)
        elif kind == "SPPF": x = R.sppf(sd, p, x, L["k"], True)
        elif kind == "C2PSA": x = R.c2psa(sd, p, x, L["n"], L["e"], True)
        elif kind == "Upsample": x = F.interpolate(x, scale_factor=2.0, mode="nearest")
        elif kind == "Concat": x = torch.cat(x, 1)
        elif kind == "Detect": x = R.detect_head(sd, p, x, L["nc"], True)
        saved.append(x)
    return saved
ref = oracle_layers()
from sy11.nn import tasks
from sy11.engine import Act
for mode in ("nograd", "grad"):
    m = tiny_model().train()
    cap = {}
    orig = tasks.BaseModel._run
    def patched(self, ec, x):
        y = []
        for mm in self.model:
            if mm.f != -1:
                x = y[mm.f] if isinstance(mm.f, int) else [x if j == -1 else y[j] for j in mm.f]
            if isinstance(mm, torch.nn.Upsample): x = tasks._upsample_run(ec, mm, x)
            else: x = mm._run(ec, x)
            cap[mm.i] = [a.data.float().permute(0,3,1,2).cpu().clone() for a in (x if isinstance(x, list) else [x])]
            y.append(x if mm.i in self.save else None)
        return x
    tasks.BaseModel._run = patched
    if mode == "nograd":
        with torch.no_grad(): m(img.cuda())
    else:
        m(img.cuda())
    tasks.BaseModel._run = orig
    print("==", mode, "save", m.save)
    for L in layers:
        r = ref[L["i"]]; r = r if isinstance(r, list) else [r]
        for a, b in zip(cap[L["i"]], r):
            print(L["i"], L["kind"], "err %.3e scale %.3e" % ((a - b).abs().max().item(), b.abs().max().item()))
