"""ORACLE — TEST INFRASTRUCTURE ONLY (never imported by the product package).

CPU restatement, in plain functional PyTorch fp32, of the reference's YOLO11 detection graph
(Conv+BN+SiLU, DWConv, Bottleneck, C3k / C3k2, SPPF, Attention / PSABlock / C2PSA, Upsample, Concat,
Detect + DFL decode).  It is driven by a flat ``state_dict`` whose keys are the reference's own
(``model.{i}.conv.weight`` ...), so reference checkpoints feed it directly.

Reference lines followed (all under /root/reference/ultralytics):
  nn/modules/conv.py:56-83     autopad / Conv          nn/modules/conv.py:687-692   DWConv
  nn/modules/block.py:65-83    DFL                     nn/modules/block.py:179-198  SPPF
  nn/modules/block.py:444-471  C2f                     nn/modules/block.py:490-504  C3
  nn/modules/block.py:713-726  Bottleneck              nn/modules/block.py:1659-1680 C3k2 / C3k
  nn/modules/block.py:1878-1933 Attention              nn/modules/block.py:1973-2007 PSABlock
  nn/modules/block.py:2100-2139 C2PSA                  nn/modules/head.py:21-172    Detect
  nn/tasks.py:161-188 graph walk, :963-1168 channel/depth scaling rules
  utils/tal.py:334-358 make_anchors / dist2bbox        utils/torch_utils.py:238-265 fuse_conv_and_bn
  utils/torch_utils.py:410-420 BN eps=1e-3, momentum=0.03
  fusion variant (config 5): nn/modules/conv.py:694-710 DDWConv, :1839-1852 WeightedSpatialAttention,
  :1854-1857,1928-1931,2087-2127 Fusion('ESChannel'), :2284-2301 GCT; nn/tasks.py:1132-1135; cfg yolo11_fusion_sand3_new.yaml

Parity status: PINNED — checked against outputs of the reference itself (tests/golden/*.npz, made by
oracle/gen_golden.py which imports /root/reference in the dev container).
"""
from __future__ import annotations

import math
import zlib

import torch
import torch.nn.functional as F

BN_EPS = 1e-3      # utils/torch_utils.py:417
BN_MOM = 0.03      # utils/torch_utils.py:418
REG_MAX = 16       # nn/modules/head.py:39

# cfg/models/11/yolo11.yaml:18-50, restated as (from, repeats, kind, args)
GRAPH = (
    (-1, 1, "Conv", (64, 3, 2)),
    (-1, 1, "Conv", (128, 3, 2)),
    (-1, 2, "C3k2", (256, False, 0.25)),
    (-1, 1, "Conv", (256, 3, 2)),
    (-1, 2, "C3k2", (512, False, 0.25)),
    (-1, 1, "Conv", (512, 3, 2)),
    (-1, 2, "C3k2", (512, True)),
    (-1, 1, "Conv", (1024, 3, 2)),
    (-1, 2, "C3k2", (1024, True)),
    (-1, 1, "SPPF", (1024, 5)),
    (-1, 2, "C2PSA", (1024,)),
    (-1, 1, "Upsample", ()),
    ((-1, 6), 1, "Concat", ()),
    (-1, 2, "C3k2", (512, False)),
    (-1, 1, "Upsample", ()),
    ((-1, 4), 1, "Concat", ()),
    (-1, 2, "C3k2", (256, False)),
    (-1, 1, "Conv", (256, 3, 2)),
    ((-1, 13), 1, "Concat", ()),
    (-1, 2, "C3k2", (512, False)),
    (-1, 1, "Conv", (512, 3, 2)),
    ((-1, 10), 1, "Concat", ()),
    (-1, 2, "C3k2", (1024, True)),
    ((16, 19, 22), 1, "Detect", ()),
)
# [depth, width, max_channels]; "t" is the tiny fixture scale of SURVEY.md appendix A
# cfg/models/11/yolo11_fusion_sand3_new.yaml:18-57 (backbone = GRAPH[:11]); DDWConv args (c2, k, s, d)
GRAPH_FUSION = GRAPH[:11] + (
    (2, 1, "DDWConv", (256, 7, 2, 2)),
    (4, 1, "Conv", (256, 1, 1)),
    (4, 1, "DDWConv", (256, 3, 2, 2)),
    (6, 1, "Conv", (256, 1, 1)),
    (10, 1, "Conv", (256, 1, 1)),
    (-1, 1, "Upsample", ()),
    ((-1, 13, 14), 1, "Fusion", ()),
    (-1, 2, "C3k2", (256, False)),
    (-1, 1, "Upsample", ()),
    ((-1, 11, 12), 1, "Fusion", ()),
    (-1, 2, "C3k2", (256, False)),
    (-1, 1, "Conv", (256, 3, 2)),
    ((-1, 18, 14), 1, "Fusion", ()),
    (-1, 2, "C3k2", (256, False)),
    (-1, 1, "Conv", (256, 3, 2)),
    ((-1, 15), 1, "Fusion", ()),
    (-1, 2, "C3k2", (256, True)),
    ((21, 24, 27), 1, "Detect", ()),
)

SCALES = {"n": (0.50, 0.25, 1024), "s": (0.50, 0.50, 1024), "m": (0.50, 1.00, 512),
          "l": (1.00, 1.00, 512), "x": (1.00, 1.50, 512), "t": (0.50, 0.125, 1024)}


def make_divisible(x, d):  # utils/ops.py:130-143
    return math.ceil(x / d) * d


def resolve_graph(scale="s", nc=80, ch=3, graph=None):
    """Channel / depth arithmetic of parse_model (nn/tasks.py:1085-1101,1136-1141)."""
    depth, width, max_ch = SCALES[scale]
    chans, layers = [], []
    for i, (f, n, kind, args) in enumerate(graph or GRAPH):
        n = max(round(n * depth), 1) if n > 1 else n
        c_in = (ch if i == 0 else chans[-1]) if f == -1 else None
        L = {"i": i, "f": f, "kind": kind}
        if kind in ("Conv", "C3k2", "SPPF", "C2PSA", "DDWConv"):
            c1 = ch if i == 0 else chans[f]
            c2 = make_divisible(min(args[0], max_ch) * width, 8)
            L.update(c1=c1, c2=c2)
            if kind == "Conv":
                L.update(k=args[1], s=args[2])
            elif kind == "DDWConv":
                L.update(k=args[1], s=args[2], d=args[3])
            elif kind == "C3k2":
                c3k = bool(args[1]) or scale in "mlx"
                e = args[2] if len(args) > 2 else 0.5
                L.update(n=n, c3k=c3k, e=e)
            elif kind == "SPPF":
                L.update(k=args[1])
            else:
                L.update(n=n, e=0.5)
        elif kind == "Upsample":
            c2 = chans[f]
        elif kind == "Concat":
            c2 = sum(chans[x] for x in f)
        elif kind == "Fusion":                 # tasks.py:1132-1135: always 'ESChannel', output width = first input's
            c2 = chans[f[0]]
        elif kind == "Detect":
            L.update(nc=nc, ch=[chans[x] for x in f])
            c2 = None
        chans.append(c2)
        L["c_out"] = c2
        layers.append(L)
    return layers


# --------------------------------------------------------------------------------------------------
# primitive ops
# --------------------------------------------------------------------------------------------------
# ---- 16-bit execution emulation ---------------------------------------------------------------------------------------------
# The reference trains under torch autocast (engine/trainer.py:261-271, :378: fp16 operands and stored activations, f32
# accumulation, f32 master weights); the product's f16 mode has the same rounding points.  Inside ``with emulate_f16():`` this
# restatement rounds at exactly those points while still computing in f32 on the CPU, so that a 16-bit run can be compared with
# something that differs from it only by summation order — not by the operand quantisation itself.  Rounding points:
#   * the input image, every conv / attention operand (filters through a straight-through rounding: master weights and their
#     gradients stay f32);
#   * every STORED activation: the raw conv output y, the BN(+SiLU)(+residual) output z (ONE rounding after the residual add:
#     the product fuses it into the BN pass), the attention output; batch statistics come from the UNROUNDED conv output
#     (the product takes them from the f32 accumulators); Detect's last 1x1 convs write f32 logits;
#   * the gradient of every stored activation (a 16-bit tensor too) — scaled by whatever loss scale the caller applies.
_EMU = {"on": False}


class _Stored16(torch.autograd.Function):
    """A tensor kept in 16-bit memory: value rounded forward, its gradient rounded backward."""

    @staticmethod
    def forward(ctx, x):
        return x.half().float()

    @staticmethod
    def backward(ctx, g):
        return g.half().float()


class _Operand16(torch.autograd.Function):
    """A 16-bit working copy of an f32 master tensor (filters): rounded value, f32 gradient (straight-through)."""

    @staticmethod
    def forward(ctx, x):
        return x.half().float()

    @staticmethod
    def backward(ctx, g):
        return g


def _st(x):
    return _Stored16.apply(x) if _EMU["on"] else x


def _op(x):
    return _Operand16.apply(x) if _EMU["on"] else x


class emulate_f16:
    """Context manager switching the restatement to the f16 execution's rounding points (see above)."""

    def __enter__(self):
        self.prev = _EMU["on"]
        _EMU["on"] = True
        return self

    def __exit__(self, *exc):
        _EMU["on"] = self.prev
        return False


def autopad(k, d=1):
    k = d * (k - 1) + 1 if d > 1 else k
    return k // 2


def conv_bn_act(sd, p, x, k=1, s=1, g=1, d=1, act=True, train=False, fused=False, res=None):
    """Conv.forward / forward_fuse.  ``p`` is the state_dict prefix of the Conv module; ``res``: a residual the caller adds
    to the result (Bottleneck / PSABlock shortcuts) — under emulate_f16 the sum is rounded once, as the fused BN pass does."""
    w = _op(sd[p + "conv.weight"])
    if fused:
        y = F.conv2d(x, w, sd[p + "conv.bias"], s, autopad(k, d), d, g)
        y = F.silu(y) if act else y
        return _st(y if res is None else res + y)
    y = F.conv2d(x, w, None, s, autopad(k, d), d, g)
    if not _EMU["on"]:
        y = F.batch_norm(y, sd[p + "bn.running_mean"], sd[p + "bn.running_var"], sd[p + "bn.weight"],
                         sd[p + "bn.bias"], train, BN_MOM, BN_EPS)
    else:
        gamma, beta = sd[p + "bn.weight"], sd[p + "bn.bias"]
        if train:                                      # statistics of the f32 accumulators, normalisation of the stored 16-bit y
            mean = y.mean((0, 2, 3))
            var = y.var((0, 2, 3), unbiased=False)
            with torch.no_grad():
                n = y.numel() / y.shape[1]
                sd[p + "bn.running_mean"].mul_(1 - BN_MOM).add_(BN_MOM * mean)
                sd[p + "bn.running_var"].mul_(1 - BN_MOM).add_(BN_MOM * var * n / max(n - 1, 1))
        else:
            mean, var = sd[p + "bn.running_mean"], sd[p + "bn.running_var"]
        scale = gamma * torch.rsqrt(var + BN_EPS)
        shift = beta - mean * scale
        y = _st(y) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    if train and (p + "bn.num_batches_tracked") in sd:
        sd[p + "bn.num_batches_tracked"] += 1
    y = F.silu(y) if act else y
    return _st(y if res is None else res + y)


def ddwconv(sd, p, x, k, s, d, train=False, fused=False):
    """DDWConv.forward (conv.py:694-710): Conv(c1, c2, k, s, g=8, d) then Conv(c2, c2, 1)."""
    y = conv_bn_act(sd, p + "conv1.", x, k, s, g=8, d=d, train=train, fused=fused)
    return conv_bn_act(sd, p + "conv2.", y, 1, 1, train=train, fused=fused)


def gct(sd, p, x, eps=1e-5):
    """GCT.forward, mode 'l2' (conv.py:2296-2301)."""
    alpha, gamma, beta = sd[p + "alpha"], sd[p + "gamma"], sd[p + "beta"]
    embedding = (x.pow(2).sum((2, 3), keepdim=True) + eps).pow(0.5) * alpha
    norm = gamma / (embedding.pow(2).mean(dim=1, keepdim=True) + eps).pow(0.5)
    return x * (1.0 + torch.tanh(embedding * norm + beta))


def weighted_spatial_attention(sd, p, x):
    """WeightedSpatialAttention(3).forward (conv.py:1850-1852)."""
    m = torch.cat([torch.mean(x, 1, keepdim=True), torch.max(x, 1, keepdim=True)[0]], 1)
    return x * torch.sigmoid(F.conv2d(m, sd[p + "cv1.weight"], None, 1, 1))


def fusion_eschannel(sd, p, xs):
    """Fusion.forward, 'ESChannel' branch (conv.py:2108-2121): chunks of GCT(cat) plus per-input spatial attention."""
    a = gct(sd, p + ("gsc2." if len(xs) == 2 else "gsc3."), torch.cat(xs, 1))
    chunks = torch.chunk(a, len(xs), dim=1)
    return _st(sum(c + weighted_spatial_attention(sd, p + "sab.", xs[i]) for i, c in enumerate(chunks)))


def bottleneck(sd, p, x, c1, c2, shortcut, k=(3, 3), e=0.5, train=False, fused=False):
    y = conv_bn_act(sd, p + "cv1.", x, k[0], 1, train=train, fused=fused)
    return conv_bn_act(sd, p + "cv2.", y, k[1], 1, train=train, fused=fused, res=x if (shortcut and c1 == c2) else None)


def c3k(sd, p, x, c, n=2, shortcut=True, train=False, fused=False):
    """C3k(c, c, 2, shortcut, g): C3 with k=3 bottlenecks at e=1.0 over hidden c/2."""
    a = conv_bn_act(sd, p + "cv1.", x, train=train, fused=fused)
    for j in range(n):
        c_ = a.shape[1]
        a = bottleneck(sd, f"{p}m.{j}.", a, c_, c_, shortcut, (3, 3), 1.0, train, fused)
    b = conv_bn_act(sd, p + "cv2.", x, train=train, fused=fused)
    return conv_bn_act(sd, p + "cv3.", torch.cat((a, b), 1), train=train, fused=fused)


def c3k2(sd, p, x, c2, n, use_c3k, e, shortcut=True, train=False, fused=False):
    c = int(c2 * e)
    y = list(conv_bn_act(sd, p + "cv1.", x, train=train, fused=fused).chunk(2, 1))
    for j in range(n):
        if use_c3k:
            y.append(c3k(sd, f"{p}m.{j}.", y[-1], c, 2, shortcut, train, fused))
        else:
            y.append(bottleneck(sd, f"{p}m.{j}.", y[-1], c, c, shortcut, (3, 3), 0.5, train, fused))
    return conv_bn_act(sd, p + "cv2.", torch.cat(y, 1), train=train, fused=fused)


def sppf(sd, p, x, k=5, train=False, fused=False):
    y = [conv_bn_act(sd, p + "cv1.", x, train=train, fused=fused)]
    for _ in range(3):
        y.append(F.max_pool2d(y[-1], k, 1, k // 2))
    return conv_bn_act(sd, p + "cv2.", torch.cat(y, 1), train=train, fused=fused)


def attention(sd, p, x, num_heads, attn_ratio=0.5, train=False, fused=False, res=None):
    B, C, H, W = x.shape
    N = H * W
    hd = C // num_heads
    kd = int(hd * attn_ratio)
    qkv = conv_bn_act(sd, p + "qkv.", x, act=False, train=train, fused=fused)
    q, k, v = qkv.view(B, num_heads, 2 * kd + hd, N).split([kd, kd, hd], dim=2)
    attn = ((q.transpose(-2, -1) @ k) * kd ** -0.5).softmax(dim=-1)
    o = _st((v @ _op(attn).transpose(-2, -1)).view(B, C, H, W))      # probabilities enter the second product as 16-bit operands
    o = conv_bn_act(sd, p + "pe.", v.reshape(B, C, H, W), 3, 1, g=C, act=False, train=train, fused=fused, res=o)
    return conv_bn_act(sd, p + "proj.", o, act=False, train=train, fused=fused, res=res)


def psablock(sd, p, x, num_heads, train=False, fused=False):
    x = attention(sd, p + "attn.", x, num_heads, 0.5, train, fused, res=x)
    y = conv_bn_act(sd, p + "ffn.0.", x, train=train, fused=fused)
    return conv_bn_act(sd, p + "ffn.1.", y, act=False, train=train, fused=fused, res=x)


def c2psa(sd, p, x, n=1, e=0.5, train=False, fused=False):
    c = int(x.shape[1] * e)
    a, b = conv_bn_act(sd, p + "cv1.", x, train=train, fused=fused).split((c, c), 1)
    for j in range(n):
        b = psablock(sd, f"{p}m.{j}.", b, c // 64, train, fused)
    return conv_bn_act(sd, p + "cv2.", torch.cat((a, b), 1), train=train, fused=fused)


def make_anchors(shapes, strides, offset=0.5, dtype=torch.float32):
    pts, st = [], []
    for (h, w), s in zip(shapes, strides):
        sx = torch.arange(w, dtype=dtype) + offset
        sy = torch.arange(h, dtype=dtype) + offset
        gy, gx = torch.meshgrid(sy, sx, indexing="ij")
        pts.append(torch.stack((gx, gy), -1).view(-1, 2))
        st.append(torch.full((h * w, 1), float(s), dtype=dtype))
    return torch.cat(pts), torch.cat(st)


def dist2bbox(dist, anchors, xywh=True, dim=-1):
    lt, rb = dist.chunk(2, dim)
    x1y1, x2y2 = anchors - lt, anchors + rb
    if xywh:
        return torch.cat(((x1y1 + x2y2) / 2, x2y2 - x1y1), dim)
    return torch.cat((x1y1, x2y2), dim)


def dfl_expect(x):
    """DFL.forward: (B, 64, A) -> (B, 4, A); channel order [side][bin]."""
    b, _, a = x.shape
    proj = torch.arange(REG_MAX, dtype=x.dtype)
    return (x.view(b, 4, REG_MAX, a).softmax(2) * proj.view(1, 1, -1, 1)).sum(2)


def detect_head(sd, p, feats, nc, train=False, fused=False):
    """Detect.forward raw maps: per level cat(cv2[i](x), cv3[i](x)); legacy=False (DWConv cls branch)."""
    outs = []
    for i, x in enumerate(feats):
        c = x.shape[1]
        b = conv_bn_act(sd, f"{p}cv2.{i}.0.", x, 3, train=train, fused=fused)
        b = conv_bn_act(sd, f"{p}cv2.{i}.1.", b, 3, train=train, fused=fused)
        b = F.conv2d(b, _op(sd[f"{p}cv2.{i}.2.weight"]), sd[f"{p}cv2.{i}.2.bias"])        # f32 logits (no output rounding)
        c3 = sd[f"{p}cv3.{i}.0.1.conv.weight"].shape[0]
        s_ = conv_bn_act(sd, f"{p}cv3.{i}.0.0.", x, 3, g=c, train=train, fused=fused)
        s_ = conv_bn_act(sd, f"{p}cv3.{i}.0.1.", s_, 1, train=train, fused=fused)
        s_ = conv_bn_act(sd, f"{p}cv3.{i}.1.0.", s_, 3, g=c3, train=train, fused=fused)
        s_ = conv_bn_act(sd, f"{p}cv3.{i}.1.1.", s_, 1, train=train, fused=fused)
        s_ = F.conv2d(s_, _op(sd[f"{p}cv3.{i}.2.weight"]), sd[f"{p}cv3.{i}.2.bias"])
        outs.append(torch.cat((b, s_), 1))
    return outs


def detect_decode(maps, strides, nc):
    """Detect._inference: (B, 4+nc, A) = cat(dist2bbox(DFL(box)) * stride, sigmoid(cls))."""
    B = maps[0].shape[0]
    no = nc + 4 * REG_MAX
    x_cat = torch.cat([m.reshape(B, no, -1) for m in maps], 2)
    anchors, st = make_anchors([m.shape[2:] for m in maps], strides)
    box, cls = x_cat.split((4 * REG_MAX, nc), 1)
    dbox = dist2bbox(dfl_expect(box), anchors.t().unsqueeze(0), xywh=True, dim=1) * st.t()
    return torch.cat((dbox, cls.sigmoid()), 1)


STRIDES = (8.0, 16.0, 32.0)


def forward(sd, layers, x, train=False, fused=False):
    """BaseModel._predict_once over the resolved graph.  train -> 3 raw maps; eval -> (y, maps)."""
    saved = []
    x = _st(x)                                     # the 16-bit stem reads the image in the compute dtype
    for L in layers:
        i, f, kind = L["i"], L["f"], L["kind"]
        p = f"model.{i}."
        if f != -1:
            x = saved[f] if isinstance(f, int) else [x if j == -1 else saved[j] for j in f]
        if kind == "Conv":
            x = conv_bn_act(sd, p, x, L["k"], L["s"], train=train, fused=fused)
        elif kind == "C3k2":
            x = c3k2(sd, p, x, L["c2"], L["n"], L["c3k"], L["e"], True, train, fused)
        elif kind == "SPPF":
            x = sppf(sd, p, x, L["k"], train, fused)
        elif kind == "C2PSA":
            x = c2psa(sd, p, x, L["n"], L["e"], train, fused)
        elif kind == "Upsample":
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
        elif kind == "Concat":
            x = torch.cat(x, 1)
        elif kind == "DDWConv":
            x = ddwconv(sd, p, x, L["k"], L["s"], L["d"], train, fused)
        elif kind == "Fusion":
            x = fusion_eschannel(sd, p, x)
        elif kind == "Detect":
            maps = detect_head(sd, p, x, L["nc"], train, fused)
            if train:
                return maps
            return detect_decode(maps, STRIDES, L["nc"]), maps
        saved.append(x)
    raise RuntimeError("graph has no Detect layer")


def fuse_state_dict(sd):
    """fuse_conv_and_bn over every Conv(+BN) in a state_dict -> new dict with conv.weight / conv.bias."""
    out = {}
    for k, v in sd.items():
        if k.endswith("bn.weight"):
            p = k[: -len("bn.weight")]
            w = sd[p + "conv.weight"]
            scale = v / torch.sqrt(sd[p + "bn.running_var"] + BN_EPS)
            out[p + "conv.weight"] = w * scale.view(-1, 1, 1, 1)
            out[p + "conv.bias"] = sd[p + "bn.bias"] - sd[p + "bn.running_mean"] * scale
        elif ".bn." in k or (k.endswith("conv.weight") and (k[: -len("conv.weight")] + "bn.weight") in sd):
            continue
        else:
            out[k] = v
    return out


# --------------------------------------------------------------------------------------------------
# closed-form deterministic tensors (shared by the golden generator and every parity test)
# --------------------------------------------------------------------------------------------------
def closed_form(key: str, shape, kind="auto"):
    """Deterministic value for a named tensor: a·sin(w·j + phase(key)) with per-kind amplitude/offset."""
    n = 1
    for s in shape:
        n *= s
    h = zlib.crc32(key.encode())
    phase = (h % 6283) / 1000.0
    freq = 0.618 + ((h >> 13) % 997) / 3000.0
    j = torch.arange(n, dtype=torch.float64)
    base = torch.sin(freq * j + phase)
    if kind == "auto":
        if key.endswith("running_var"):
            kind = "var"
        elif key.endswith("running_mean"):
            kind = "mean"
        elif key.endswith("bn.weight"):
            kind = "gamma"
        elif key.endswith("bias"):
            kind = "bias"
        else:
            kind = "weight"
    if kind == "weight":
        fan_in = max(n // max(shape[0], 1), 1)
        v = base * math.sqrt(3.0 / fan_in)
    elif kind == "gamma":
        v = 1.0 + 0.25 * base
    elif kind == "bias":
        v = 0.1 * base
    elif kind == "mean":
        v = 0.1 * base
    elif kind == "var":
        v = 1.0 + 0.5 * base * base
    elif kind == "input":
        v = 0.5 + 0.5 * base           # in [0, 1]
    elif kind == "signed":
        v = base
    else:
        raise ValueError(kind)
    return v.to(torch.float32).view(*shape)


def closed_form_state_dict(template: dict):
    """Overwrite every float entry of a state_dict (SURVEY §8g gotcha 2) except the frozen DFL weight."""
    out = {}
    for k, v in template.items():
        if not v.dtype.is_floating_point or ".dfl." in k:
            out[k] = v.clone()
        else:
            out[k] = closed_form(k, tuple(v.shape))
    return out


def seeded_state_dict(template: dict, seed: int = 0):
    """Seeded-random (CPU generator) values for every float entry except the frozen DFL weight.

    Used for END-TO-END fixtures: the sinusoidal closed form above yields near-degenerate channels, which makes
    train-mode BatchNorm over tiny batches ill-conditioned (fp32 vs fp64 of the same graph differ by 3e-2), so no
    implementation could be compared at 1e-3.  He-style random filters keep fp32-vs-fp64 at ~1e-4."""
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, v in template.items():
        if not v.dtype.is_floating_point or ".dfl." in k:
            out[k] = v.clone()
        elif k.endswith("running_var"):
            out[k] = 0.5 + torch.rand(v.shape, generator=g)
        elif k.endswith("running_mean"):
            out[k] = 0.1 * torch.randn(v.shape, generator=g)
        elif k.endswith("bn.weight"):
            out[k] = 1 + 0.2 * torch.randn(v.shape, generator=g)
        elif k.endswith("bias"):
            out[k] = 0.1 * torch.randn(v.shape, generator=g)
        elif k.endswith(".alpha"):              # GCT parameters (fusion variant): alpha ~ 1, gamma / beta small but alive
            out[k] = 1 + 0.2 * torch.randn(v.shape, generator=g)
        elif k.endswith(".gamma") or k.endswith(".beta"):
            out[k] = 0.3 * torch.randn(v.shape, generator=g)
        else:
            out[k] = torch.randn(v.shape, generator=g) * math.sqrt(2.0 / v[0].numel())
    return out


def pattern_state_dict(template: dict):
    """Low-entropy, f16-exact, deterministic values (multiples of 1/8 from a short repeating cycle, per-key rotation):
    used for the checkpoint fixture so the pickled file compresses to a few KB.  BN variances stay positive."""
    out = {}
    cyc = torch.tensor([0.25, -0.125, 0.5, 0.0, -0.375, 0.125, -0.25, 0.375, -0.5])
    for k, v in template.items():
        if not v.dtype.is_floating_point or ".dfl." in k:
            out[k] = v.clone()
            continue
        n = v.numel()
        r = zlib.crc32(k.encode()) % 9
        base = cyc[(torch.arange(n) + r) % 9].view(v.shape)
        if k.endswith("running_var"):
            out[k] = 1.0 + base.abs()
        elif k.endswith("bn.weight"):
            out[k] = 1.0 + 0.5 * base
        elif k.endswith("conv.weight") or k.endswith(".weight"):
            out[k] = base * (2.0 ** -round(math.log2(max(v[0].numel(), 1)) / 2))      # ~ fan-in scaling, still f16-exact
        else:
            out[k] = 0.25 * base
    return out


def seeded_image(shape, seed: int = 5):
    return torch.rand(shape, generator=torch.Generator().manual_seed(seed))


def empty_state_dict(layers):
    """Allocate a zero state_dict with the reference's keys and shapes for a resolved graph."""
    sd = {}

    def conv(p, c1, c2, k=1, g=1):
        sd[p + "conv.weight"] = torch.zeros(c2, c1 // g, k, k)
        for nme in ("weight", "bias", "running_mean", "running_var"):
            sd[p + "bn." + nme] = torch.zeros(c2)
        sd[p + "bn.num_batches_tracked"] = torch.zeros((), dtype=torch.long)

    def bneck(p, c1, c2, e):
        c_ = int(c2 * e)
        conv(p + "cv1.", c1, c_, 3)
        conv(p + "cv2.", c_, c2, 3)

    for L in layers:
        p = f"model.{L['i']}."
        kind = L["kind"]
        if kind == "Conv":
            conv(p, L["c1"], L["c2"], L["k"])
        elif kind == "C3k2":
            c = int(L["c2"] * L["e"])
            conv(p + "cv1.", L["c1"], 2 * c)
            conv(p + "cv2.", (2 + L["n"]) * c, L["c2"])
            for j in range(L["n"]):
                q = f"{p}m.{j}."
                if L["c3k"]:
                    c_ = int(c * 0.5)
                    conv(q + "cv1.", c, c_)
                    conv(q + "cv2.", c, c_)
                    conv(q + "cv3.", 2 * c_, c)
                    for t in range(2):
                        bneck(f"{q}m.{t}.", c_, c_, 1.0)
                else:
                    bneck(q, c, c, 0.5)
        elif kind == "DDWConv":
            conv(p + "conv1.", L["c1"], L["c2"], L["k"], g=8)
            conv(p + "conv2.", L["c2"], L["c2"])
        elif kind == "Fusion":                  # Fusion(inc_list, 'ESChannel', c1=128): widths hard-coded (conv.py:1855)
            sd[p + "sab.cv1.weight"] = torch.zeros(1, 2, 3, 3)
            for nme, mult in (("gsc2.", 2), ("gsc3.", 3)):
                for q in ("alpha", "gamma", "beta"):
                    sd[p + nme + q] = torch.zeros(1, 128 * mult, 1, 1)
        elif kind == "SPPF":
            c_ = L["c1"] // 2
            conv(p + "cv1.", L["c1"], c_)
            conv(p + "cv2.", 4 * c_, L["c2"])
        elif kind == "C2PSA":
            c = int(L["c1"] * L["e"])
            conv(p + "cv1.", L["c1"], 2 * c)
            conv(p + "cv2.", 2 * c, L["c1"])
            for j in range(L["n"]):
                q = f"{p}m.{j}."
                nh = c // 64
                kd = int(c // nh * 0.5)
                conv(q + "attn.qkv.", c, c + 2 * nh * kd)
                conv(q + "attn.proj.", c, c)
                conv(q + "attn.pe.", c, c, 3, g=c)
                conv(q + "ffn.0.", c, 2 * c)
                conv(q + "ffn.1.", 2 * c, c)
        elif kind == "Detect":
            nc, ch = L["nc"], L["ch"]
            c2 = max(16, ch[0] // 4, 4 * REG_MAX)
            c3 = max(ch[0], min(nc, 100))
            for i, x in enumerate(ch):
                conv(f"{p}cv2.{i}.0.", x, c2, 3)
                conv(f"{p}cv2.{i}.1.", c2, c2, 3)
                sd[f"{p}cv2.{i}.2.weight"] = torch.zeros(4 * REG_MAX, c2, 1, 1)
                sd[f"{p}cv2.{i}.2.bias"] = torch.zeros(4 * REG_MAX)
                conv(f"{p}cv3.{i}.0.0.", x, x, 3, g=x)
                conv(f"{p}cv3.{i}.0.1.", x, c3)
                conv(f"{p}cv3.{i}.1.0.", c3, c3, 3, g=c3)
                conv(f"{p}cv3.{i}.1.1.", c3, c3)
                sd[f"{p}cv3.{i}.2.weight"] = torch.zeros(nc, c3, 1, 1)
                sd[f"{p}cv3.{i}.2.bias"] = torch.zeros(nc)
            sd[p + "dfl.conv.weight"] = torch.arange(REG_MAX, dtype=torch.float32).view(1, REG_MAX, 1, 1)
    return sd
