"""GPU: the hot path at BASELINE.json's full sizes (yolo11s, 640 x 640, batch 64) through size-independent properties —
the oracle cannot run these shapes in seconds, the identities below need no reference values:

  * adjointness: <conv(x; w), dy> = <x, dgrad(dy; w)> = <w, wgrad(x, dy)>  (one scalar ties the three conv kernels of a
    layer together; evaluated in f64 on the device from the kernels' outputs)
  * linearity of the forward kernel in x
  * BatchNorm + SiLU apply: per-channel mean / variance of the normalised tensor
  * the whole model: the training loss and gradients are invariant under a permutation of the batch (train-mode BN
    statistics are permutation invariant), and two replays of the captured graph on the same batch agree.
The tile autotuner is off in the test suite (tests/conftest.py): the heuristic configurations — incl. the halo-tiled 3x3 kernel — are what is
checked here; every configuration is forced on its own in test_kernels_gpu.py."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"

# (IH, IW, C, N, k, s, groups) — layers of yolo11s at 640 x 640 (tools/conv_sweep.py LAYERS), batch 64
LAYERS = [(320, 320, 32, 64, 3, 2, 1), (160, 160, 128, 128, 3, 2, 1), (80, 80, 512, 128, 1, 1, 1), (80, 80, 64, 64, 3, 1, 1),
          (40, 40, 768, 256, 1, 1, 1), (40, 40, 256, 512, 3, 2, 1), (20, 20, 128, 128, 3, 1, 1), (20, 20, 1024, 512, 1, 1, 1),
          (80, 80, 128, 128, 3, 1, 128), (160, 160, 16, 32, 3, 1, 1)]


def dot(a, b):
    return (a.double() * b.double()).sum().item()


@pytest.mark.parametrize("dtype,tol", [(torch.float16, 4e-3), (torch.float32, 2e-4)])
@pytest.mark.parametrize("layer", LAYERS, ids=[f"{l[0]}x{l[1]}_{l[2]}to{l[3]}_k{l[4]}s{l[5]}g{l[6]}" for l in LAYERS])
def test_conv_adjoint_identities_full_size(layer, dtype, tol):
    from sy11 import ops
    H, W, C, N, k, s, g = layer
    B, p = 64, k // 2
    if dtype == torch.float32 and H >= 160:
        B = 16                                                   # keep the f32 copies of the two largest maps modest
    OH, OW = ops.conv_out_hw(H, W, k, s, p)
    gen = torch.Generator(device=DEV).manual_seed(H * 1000 + C)
    x = torch.randn(B, H, W, C, device=DEV, generator=gen).to(dtype)
    w = (torch.randn(N, k, k, C // g, device=DEV, generator=gen) / math.sqrt(C // g * k * k)).to(dtype)
    dy = torch.randn(B, OH, OW, N, device=DEV, generator=gen).to(dtype)
    y = torch.empty(B, OH, OW, N, device=DEV, dtype=dtype)
    dx = torch.zeros_like(x)
    dw = torch.zeros(N, k, k, C // g, device=DEV)
    st = torch.zeros(2, 32, N, device=DEV)
    ops.conv2d_fwd(x, w, y, k, s, p, groups=g, stats=(st[0], st[1]))
    ops.conv2d_dgrad(dy, ops.weight_transpose(w) if g == 1 else w, dx, (B, OH, OW, N), k, s, p, groups=g, accumulate=(s > 1 and g == 1))
    ops.conv2d_wgrad(x, dy, dw, k, s, p, groups=g)
    a, b, c = dot(y, dy), dot(x, dx), dot(w, dw)
    # scale of the sums: sqrt(#terms) * typical |term|; the three numbers are the same bilinear form
    scale = math.sqrt(B * OH * OW * N) * (y.float().std().item() + 1e-6)
    assert abs(a - b) <= tol * scale * 8 and abs(a - c) <= tol * scale * 8, (a, b, c, scale)
    # BN statistics epilogue = column sums of what was stored (f32 accumulators vs the rounded output: dtype tolerance)
    cs = y.double().sum((0, 1, 2))
    assert torch.allclose(st[0].sum(0).double(), cs, rtol=0, atol=tol * 16 * math.sqrt(B * OH * OW) * y.float().abs().max().item())


def test_conv_forward_is_linear_full_size():
    from sy11 import ops
    B, H, W, C, N, k, s = 64, 80, 80, 128, 128, 3, 1
    gen = torch.Generator(device=DEV).manual_seed(3)
    x1 = torch.randn(B, H, W, C, device=DEV, generator=gen)
    x2 = torch.randn(B, H, W, C, device=DEV, generator=gen)
    w = torch.randn(N, k, k, C, device=DEV, generator=gen) / math.sqrt(C * k * k)
    ys = []
    for x in (x1, x2, 0.5 * x1 - 2.0 * x2):
        y = torch.empty(B, H, W, N, device=DEV)
        ops.conv2d_fwd(x.contiguous(), w, y, k, s, 1)
        ys.append(y)
    lin = 0.5 * ys[0] - 2.0 * ys[1]
    assert (ys[2] - lin).abs().max().item() <= 2e-4 * lin.abs().max().item()


def test_bn_silu_apply_normalises_full_size():
    from sy11 import ops
    B, H, W, C = 64, 160, 160, 64
    gen = torch.Generator(device=DEV).manual_seed(5)
    y = (torch.randn(B, H, W, C, device=DEV, generator=gen) * 3 + 1.5).half()
    m = y.float().mean((0, 1, 2))
    v = y.float().var((0, 1, 2), unbiased=False)
    gamma = torch.linspace(0.5, 2.0, C, device=DEV)
    beta = torch.linspace(-1, 1, C, device=DEV)
    rstd = (v + 1e-3).rsqrt()
    scale, shift = gamma * rstd, beta - m * gamma * rstd
    z = torch.empty_like(y)
    ops.bn_act_fwd(y, scale, shift, z, silu=False)
    zm, zv = z.float().mean((0, 1, 2)), z.float().var((0, 1, 2), unbiased=False)
    assert torch.allclose(zm, beta, atol=5e-3) and torch.allclose(zv, gamma * gamma * v / (v + 1e-3), rtol=5e-3, atol=1e-3)
    zs = torch.empty_like(y)
    ops.bn_act_fwd(y, scale, shift, zs, silu=True)
    assert torch.allclose(zs.float(), torch.nn.functional.silu(z.float()), atol=4e-3, rtol=4e-3)


def _trainer(seed=1):
    from oracle import yolo11_ref as R
    from sy11.engine.trainer import DetectionTrainer
    from sy11.nn.tasks import DetectionModel
    torch.manual_seed(seed)
    m = DetectionModel("yolo11s.yaml", nc=80, verbose=False)
    if seed == 1:                                # seeded He-style fixture weights; otherwise the constructor's own initialisation
        m.load_state_dict(R.seeded_state_dict(R.empty_state_dict(R.resolve_graph("s", nc=80)), seed=seed))
    return DetectionTrainer(m, batch_size=64, device=DEV, overrides={"amp": True}, graphs=True)


def test_model_loss_and_grads_are_batch_permutation_invariant_full_size():
    """yolo11s, 64 x 3 x 640 x 640, f16: forward + loss + backward on a batch and on a permutation of it."""
    from sy11.engine import GradStore  # noqa: F401
    B = 64
    gen = torch.Generator().manual_seed(0)
    img = torch.rand(B, 3, 640, 640, generator=gen).to(DEV)
    nl = 200
    bi = torch.randint(0, B, (nl,), generator=gen).float().sort().values
    cls = torch.randint(0, 80, (nl, 1), generator=gen).float()
    box = torch.cat((0.2 + 0.6 * torch.rand(nl, 2, generator=gen), 0.05 + 0.3 * torch.rand(nl, 2, generator=gen)), 1)
    perm = torch.randperm(B, generator=gen)
    inv = torch.empty_like(perm)
    inv[perm] = torch.arange(B)
    tr = _trainer(seed=3)                                     # constructor initialisation: a well-conditioned loss
    tr.model.train()

    def run(images, batch_idx):
        store = tr.model.__dict__["_sy11_grads"]
        store.flat.zero_()
        loss, items = tr.model({"img": images, "batch_idx": batch_idx.to(DEV), "cls": cls.to(DEV), "bboxes": box.to(DEV)})
        (loss * 256.0).backward()                             # a fixed loss scale, as AMP applies: keeps f16 gradients off the underflow edge
        return loss.item(), items.clone(), store.flat.clone()

    l1, i1, g1 = run(img, bi)
    l2, i2, g2 = run(img[perm.to(DEV)].contiguous(), inv[bi.long()].float())   # image j moves to slot inv[j]
    # f16 operands + f32 atomics: only the summation order differs between the two runs
    assert math.isfinite(l1) and abs(l1 - l2) <= 5e-3 * abs(l1), (l1, l2)
    assert torch.allclose(i1, i2, rtol=5e-3, atol=1e-4)
    assert torch.isfinite(g1).all() and torch.isfinite(g2).all()
    cos = torch.dot(g1, g2).item() / (g1.norm().item() * g2.norm().item() + 1e-30)
    assert cos > 0.97, cos                                    # f16 gradients: per-tensor noise of a few % is inherent


def test_graph_replay_is_repeatable_full_size():
    tr = _trainer(seed=2)
    gen = torch.Generator().manual_seed(1)
    batch = {"img": torch.rand(64, 3, 640, 640, generator=gen).to(DEV), "batch_idx": torch.arange(64.0).to(DEV),
             "cls": torch.randint(0, 80, (64, 1), generator=gen).float().to(DEV),
             "bboxes": torch.cat((0.3 + 0.4 * torch.rand(64, 2, generator=gen), 0.1 + 0.2 * torch.rand(64, 2, generator=gen)), 1).to(DEV)}
    tr.args.lr0 = 0.0
    for g in tr.optimizer.param_groups:
        g["lr"] = 0.0
        g["initial_lr"] = 0.0
        g["weight_decay"] = 0.0
    losses = [tr.train_step(dict(batch))[0].item() for _ in range(6)]     # 2 eager + capture + replays; lr 0: same weights
    assert all(math.isfinite(l) for l in losses)
    assert "_sy11_graph_cfg" in tr.model.__dict__ and len(tr.model.__dict__["_sy11_graph_cfg"]["entries"]) == 1
    # BN running statistics move, the training-mode loss does not depend on them: replays must agree with the eager steps
    # (f32 atomics in the statistics / gradient sums reorder between runs; nothing else may differ)
    assert max(losses) - min(losses) <= 3e-3 * abs(losses[0]), losses


# ---- configs[4]: the fusion variant's own kernels at full size (yolo11s_fusion_sand3_new @ 640 x 640, batch 64)
@pytest.mark.parametrize("H,C,N,k", [(160, 128, 128, 7), (80, 256, 128, 3)], ids=["layer11_k7_d2_g8", "layer13_k3_d2_g8"])
def test_ddwconv_grouped_dilated_adjoint_full_size(H, C, N, k):
    """DDWConv.conv1 = Conv(c1, c2, k, s=2, g=8, d=2) (conv.py:694-710; cfg yolo11_fusion_sand3_new.yaml:33-37) at batch 64:
    <conv(x; w), dy> = <x, dgrad(dy; w)> = <w, wgrad(x, dy)> through the grouped / dilated paths; stride and dilation share a factor,
    so half of the input pixels receive no tap: dgrad must leave them exactly zero."""
    from sy11 import ops
    B, s, d, g, dtype, tol = 64, 2, 2, 8, torch.float16, 4e-3
    p = (d * (k - 1) + 1) // 2
    OH, OW = ops.conv_out_hw(H, H, k, s, p, d)
    gen = torch.Generator(device=DEV).manual_seed(H + k)
    x = torch.randn(B, H, H, C, device=DEV, generator=gen).to(dtype)
    w = (torch.randn(N, k, k, C // g, device=DEV, generator=gen) / math.sqrt(C // g * k * k)).to(dtype)
    dy = torch.randn(B, OH, OW, N, device=DEV, generator=gen).to(dtype)
    y = torch.empty(B, OH, OW, N, device=DEV, dtype=dtype)
    dx = torch.zeros_like(x)
    dw = torch.zeros(N, k, k, C // g, device=DEV)
    ops.conv2d_fwd(x, w, y, k, s, p, d, g)
    assert ops.dgrad_leaves_holes(k, s, p, d)
    ops.conv2d_dgrad(dy, ops.weight_transpose(w, g), dx, (B, OH, OW, N), k, s, p, d, g, accumulate=True)
    ops.conv2d_wgrad(x, dy, dw, k, s, p, d, g)
    a, b, c = dot(y, dy), dot(x, dx), dot(w, dw)
    scale = math.sqrt(B * OH * OW * N) * (y.float().std().item() + 1e-6)
    assert abs(a - b) <= tol * scale * 8 and abs(a - c) <= tol * scale * 8, (a, b, c, scale)
    # taps land on input pixels of one parity only (iy = 2*oy - p + 2*r): the other rows / columns stay exactly zero
    par = (p % 2)
    assert float(dx[:, (1 - par)::2].abs().max()) == 0.0 and float(dx[:, :, (1 - par)::2].abs().max()) == 0.0
    assert float(dx[:, par::2, par::2].abs().max()) > 0.0


def test_fusion_eschannel_directional_derivative_full_size():
    """Fusion([128, 128, 128], 'ESChannel') (conv.py:2087-2127) on three 64 x 128 x 80 x 80 maps, f32: the backward kernels
    (fusion_bwd_reduce / sab_map_bwd / gct_gate_bwd / fusion_bwd_apply) against a central finite difference of L = 0.5 * sum(out^2) along a random
    direction — (L(x + e v) - L(x - e v)) / 2e = sum_i <dL/dx_i, v_i> — and batch-permutation equivariance of the forward."""
    from sy11.nn.modules import Fusion
    torch.manual_seed(0)
    m = Fusion([128, 128, 128], "ESChannel").to(DEV).train()
    with torch.no_grad():
        for p_ in m.parameters():
            p_.add_(0.2 * torch.randn_like(p_))
    gen = torch.Generator(device=DEV).manual_seed(9)
    xs = [torch.randn(64, 128, 80, 80, device=DEV, generator=gen).contiguous(memory_format=torch.channels_last).requires_grad_(True) for _ in range(3)]
    vs = [torch.randn(64, 128, 80, 80, device=DEV, generator=gen).contiguous(memory_format=torch.channels_last) for _ in range(3)]
    out = m(xs)
    (0.5 * out * out).sum().backward()                         # L = 0.5 sum out^2: a coherent signal (a random cotangent drowns it in noise)
    lhs = sum(dot(x.grad, v) for x, v in zip(xs, vs))
    eps = 1e-2
    with torch.no_grad():
        fp = m([(x + eps * v) for x, v in zip(xs, vs)])
        fm = m([(x - eps * v) for x, v in zip(xs, vs)])
        rhs = (0.5 * dot(fp, fp) - 0.5 * dot(fm, fm)) / (2 * eps)
        perm = torch.randperm(64, device=DEV, generator=gen)
        outp = m([x[perm].contiguous(memory_format=torch.channels_last) for x in xs])
    # central difference: O(eps^2) truncation plus the kinks of the channel-max inside the spatial attention (a few pixels change
    # their arg-max along the direction)
    assert abs(lhs - rhs) <= 2e-2 * abs(rhs), (lhs, rhs)
    assert (outp - out[perm]).abs().max().item() <= 1e-4 * out.abs().max().item()
