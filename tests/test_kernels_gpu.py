"""GPU parity of every libsy11 kernel (through the C-ABI) against a CPU fp32 reference of the same op.

Tolerances: f32 kernels use exact-f32 MFMA (k-ordered fmaf chains) -> 2e-5 relative to the output scale;
f16/bf16 inputs are rounded once (operands) with f32 accumulation -> 4e-3 / 3e-2 of the output scale.
Index outputs (max-pool argmax, NMS keep) are compared bit-exactly.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"
TOL = {torch.float32: 2e-5, torch.float16: 4e-3, torch.bfloat16: 3e-2}


def ops():
    from sy11 import ops as o
    return o


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def nhwc(t_nchw, dtype, pad_c=0):
    """NCHW cpu f32 -> NHWC device view (optionally inside a wider buffer to exercise ld > C)."""
    B, Cn, H, W = t_nchw.shape
    buf = torch.zeros(B, H, W, Cn + pad_c, dtype=dtype, device=DEV)
    v = buf[..., pad_c // 2: pad_c // 2 + Cn] if pad_c else buf
    v.copy_(t_nchw.permute(0, 2, 3, 1).to(DEV, dtype))
    return v


def to_nchw(v):
    return v.float().cpu().permute(0, 3, 1, 2).contiguous()


def close(got, ref, dtype, what, mult=1.0):
    scale = max(ref.abs().max().item(), 1e-6)
    err = (got - ref).abs().max().item()
    assert err <= TOL[dtype] * mult * scale, f"{what}: max err {err:.3e} vs scale {scale:.3e} ({dtype})"


def q(t, dtype):
    """Round a cpu f32 tensor through dtype (what the kernel's operands see)."""
    return t.to(dtype).float()


CONV_CASES = [
    # B, C, H, W, N, k, s, pad_c
    (2, 32, 9, 11, 64, 1, 1, 0),
    (2, 64, 8, 8, 48, 3, 1, 32),
    (1, 16, 13, 10, 32, 3, 2, 0),
    (3, 96, 6, 7, 160, 3, 1, 16),
    (2, 128, 8, 8, 256, 1, 1, 0),
    (2, 64, 20, 20, 16, 3, 2, 64),
    (1, 8, 5, 5, 8, 3, 1, 0),
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16, torch.bfloat16])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd_dgrad_wgrad(case, dtype):
    o = ops()
    B, Cn, H, W, N, k, s, pad_c = case
    if dtype != torch.float32 and (Cn % 8 or N % 8):
        pytest.skip("16-bit path needs channels % 8 == 0")
    p = k // 2
    x = rnd(B, Cn, H, W, seed=1)
    w = rnd(N, Cn, k, k, seed=2, scale=1.0 / math.sqrt(Cn * k * k))
    bias = rnd(N, seed=3)
    xq, wq = q(x, dtype), q(w, dtype)
    OH, OW = o.conv_out_hw(H, W, k, s, p)
    xv = nhwc(x, dtype, pad_c)
    wk = wq.permute(0, 2, 3, 1).contiguous().to(DEV, dtype)
    # forward: raw + stats
    y = torch.empty(B, OH, OW, N, dtype=dtype, device=DEV)
    ssum = torch.zeros(N, device=DEV)
    ssq = torch.zeros(N, device=DEV)
    o.conv2d_fwd(xv, wk, y, k, s, p, stats=(ssum, ssq))
    ref = F.conv2d(xq, wq, None, s, p)
    close(to_nchw(y), ref, dtype, "conv fwd")
    close(ssum.cpu(), ref.sum((0, 2, 3)), dtype, "conv stats sum", mult=4 * math.sqrt(B * OH * OW))
    close(ssq.cpu(), (ref * ref).sum((0, 2, 3)), dtype, "conv stats sq", mult=4)
    # forward: bias + silu epilogue into a channel slice, f32 output flag
    ybuf = torch.zeros(B, OH, OW, N + 16, dtype=dtype, device=DEV)
    o.conv2d_fwd(xv, wk, ybuf[..., 8:8 + N], k, s, p, bias=bias.to(DEV), silu=True)
    close(to_nchw(ybuf[..., 8:8 + N]), F.silu(ref + bias.view(1, -1, 1, 1)), dtype, "conv fwd bias+silu slice")
    assert ybuf[..., :8].abs().max().item() == 0 and ybuf[..., 8 + N:].abs().max().item() == 0
    y32 = torch.empty(B, OH, OW, N, dtype=torch.float32, device=DEV)
    o.conv2d_fwd(xv, wk, y32, k, s, p, bias=bias.to(DEV), out_f32=True)
    close(to_nchw(y32), ref + bias.view(1, -1, 1, 1), dtype, "conv fwd f32 out")
    # dgrad (+ accumulate)
    dy = rnd(B, N, OH, OW, seed=4)
    dyq = q(dy, dtype)
    dyv = nhwc(dy, dtype, 16 if dtype != torch.float32 else 8)
    wt = o.weight_transpose(wk)
    assert torch.equal(wt.cpu(), wk.cpu().permute(3, 1, 2, 0).contiguous())
    dx = torch.zeros(B, H, W, Cn, dtype=dtype, device=DEV)
    o.conv2d_dgrad(dyv, wt, dx, (B, OH, OW, N), k, s, p, accumulate=(s > 1))
    ref_dx = torch.nn.grad.conv2d_input((B, Cn, H, W), wq, dyq, s, p)
    close(to_nchw(dx), ref_dx, dtype, "conv dgrad")
    o.conv2d_dgrad(dyv, wt, dx, (B, OH, OW, N), k, s, p, accumulate=True)
    close(to_nchw(dx), 2 * ref_dx, dtype, "conv dgrad accumulate", mult=2)
    # wgrad (accumulates into f32)
    dw = torch.zeros(N, k, k, Cn, dtype=torch.float32, device=DEV)
    o.conv2d_wgrad(xv, dyv, dw, k, s, p)
    ref_dw = torch.nn.grad.conv2d_weight(xq, (N, Cn, k, k), dyq, s, p)
    close(dw.cpu().permute(0, 3, 1, 2), ref_dw, torch.float32, "conv wgrad", mult=8)


CFG_CASES = [
    # B, C, H, W, N, k, s : every tile configuration must be right on its own, not just the one the tuner would pick
    (2, 64, 24, 20, 160, 3, 1),      # 3x3, K = 576 (deep rings legal), N not a multiple of the 128 / 64 tiles
    (3, 256, 12, 12, 96, 1, 1),      # 1x1, K = 256
    (2, 128, 18, 18, 64, 3, 2),      # stride 2: dgrad = 4 parity launches with 1 / 2 / 2 / 4 taps
    (4, 32, 40, 40, 32, 1, 1),       # K = 32: shorter than every deep ring (those configurations must refuse, not misbehave)
    # shapes the halo-tiled 3x3 kernel (cfg 15 / 16) takes: output width a multiple of 16, or exactly 20 / 40
    (2, 64, 32, 32, 96, 3, 2),       # stride 2 -> 16x16, N = 96 (partial channel tile), de-interleaved patch columns
    (1, 32, 40, 40, 32, 3, 1),       # whole-row tiles of a 40-wide map (3 x 40), one channel slab, N = 32
    (2, 96, 16, 48, 64, 3, 1),       # 8 x 16 tiles, three slabs, non-square map
    (1, 64, 80, 80, 40, 3, 2),       # stride 2 -> 40x40 (3 x 40 tiles over an 81-column patch)
    (1, 32, 31, 39, 64, 3, 2),       # stride 2 from odd input sizes -> 16 x 20: the patch runs past the right / bottom border
    (3, 128, 20, 20, 128, 3, 1),     # 6 x 20 tiles on the 20x20 maps of the model (partial last tile row)
    # shapes the few-channel kernel (igemm cfg 19) takes, forward and input gradient: C, N in {16, 32}, ragged 4 x 64 / 8 x 32 tiles
    (2, 16, 12, 70, 32, 3, 1),
    (1, 32, 9, 130, 16, 3, 1),
    (2, 32, 11, 96, 32, 3, 1),       # width a multiple of 32 but not of 64: 8 x 32 tiles
    (1, 16, 20, 160, 16, 3, 1),
    # shapes the patch filter-gradient kernel (wgrad cfg 12-15) takes: 3x3 stride 1, 64-blocks, 80 pixels per stage
    (2, 64, 40, 40, 64, 3, 1),       # two rows of a 40-wide map per stage
    (1, 128, 6, 80, 64, 3, 1),       # two rows of an 80-wide map per stage (160 pixels), two channel blocks
    (1, 64, 5, 80, 64, 3, 1),        # an odd row count: one row (80 pixels) per stage
    (1, 32, 4, 80, 64, 3, 1),        # 32-channel block: two waves share the k-steps of a 32 x 32 block
    (1, 64, 3, 80, 32, 3, 1),        # 32-filter block, 80-pixel stages
    (1, 16, 2, 160, 32, 3, 1),       # 16 of 32 channels real, one block: four waves share its k-steps
    (1, 32, 2, 160, 16, 3, 1),       # 16 of 32 filters real
    (1, 64, 4, 160, 128, 3, 1),      # half a row of a 160-wide map per stage, two filter blocks
    (2, 256, 20, 20, 288, 1, 1),     # wide layers (N, C >= 256): 256 x 128 tiles forward AND input gradient, partial tiles in both directions
    (1, 256, 26, 22, 256, 3, 2),     # the same through 9 taps at stride 2 (the input gradient's parity launches have K = 256 .. 1024)
]


@pytest.mark.parametrize("case", CFG_CASES)
def test_every_igemm_and_wgrad_tile_configuration(case):
    """Force each igemm (fwd / dgrad) and wgrad configuration through sy11_set_option and compare with the fp32 CPU conv.
    A configuration that is not legal for a problem falls back to the heuristic pick (so the result must still be right)."""
    from sy11 import _lib
    o = ops()
    dtype = torch.float16
    B, Cn, H, W, N, k, s = case
    p = k // 2
    x = rnd(B, Cn, H, W, seed=11)
    w = rnd(N, Cn, k, k, seed=12, scale=1.0 / math.sqrt(Cn * k * k))
    dy = rnd(B, N, *o.conv_out_hw(H, W, k, s, p), seed=13)
    xq, wq, dyq = q(x, dtype), q(w, dtype), q(dy, dtype)
    OH, OW = o.conv_out_hw(H, W, k, s, p)
    xv, dyv = nhwc(x, dtype), nhwc(dy, dtype)
    wk = wq.permute(0, 2, 3, 1).contiguous().to(DEV, dtype)
    wt = o.weight_transpose(wk)
    ref = F.conv2d(xq, wq, None, s, p)
    ref_dx = torch.nn.grad.conv2d_input((B, Cn, H, W), wq, dyq, s, p)
    ref_dw = torch.nn.grad.conv2d_weight(xq, (N, Cn, k, k), dyq, s, p)
    tune0 = _lib.get_option("tune")
    try:
        _lib.set_option("tune", 0)
        for cfg in range(26):
            _lib.set_option("igemm_cfg", cfg)
            y = torch.empty(B, OH, OW, N, dtype=dtype, device=DEV)
            ssum, ssq = torch.zeros(N, device=DEV), torch.zeros(N, device=DEV)
            o.conv2d_fwd(xv, wk, y, k, s, p, stats=(ssum, ssq))
            close(to_nchw(y), ref, dtype, f"fwd cfg {cfg}")
            close(ssum.cpu(), ref.sum((0, 2, 3)), dtype, f"stats cfg {cfg}", mult=4 * math.sqrt(B * OH * OW))
            dx = torch.zeros(B, H, W, Cn, dtype=dtype, device=DEV)
            o.conv2d_dgrad(dyv, wt, dx, (B, OH, OW, N), k, s, p, accumulate=(s > 1))
            close(to_nchw(dx), ref_dx, dtype, f"dgrad cfg {cfg}")
            o.conv2d_dgrad(dyv, wt, dx, (B, OH, OW, N), k, s, p, accumulate=True)
            close(to_nchw(dx), 2 * ref_dx, dtype, f"dgrad accumulate cfg {cfg}", mult=2)
        _lib.set_option("igemm_cfg", -1)
        for cfg in range(20):
            _lib.set_option("wgrad_cfg", cfg)
            dw = torch.zeros(N, k, k, Cn, dtype=torch.float32, device=DEV)
            o.conv2d_wgrad(xv, dyv, dw, k, s, p)
            close(dw.cpu().permute(0, 3, 1, 2), ref_dw, torch.float32, f"wgrad cfg {cfg}", mult=8)
    finally:
        _lib.set_option("igemm_cfg", -1)
        _lib.set_option("wgrad_cfg", -1)
        _lib.set_option("tune", tune0)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("case", [(2, 16, 20, 96, 32), (1, 32, 12, 160, 16), (2, 32, 9, 70, 24)])
def test_few_channel_conv_epilogues_and_bf16(case, dtype):
    """igemm configuration 19 (conv3x3s.hip) in both 16-bit types with every epilogue it takes: BN statistics (training), bias +
    SiLU (the fused inference conv), accumulate (an input gradient added to an existing one), and into a channel slice of a wider
    tensor (concat by pointer: y_ld > N, x_ld > C)."""
    from sy11 import _lib
    o = ops()
    B, Cn, H, W, N = case
    x = rnd(B, Cn, H, W, seed=41)
    w = rnd(N, Cn, 3, 3, seed=42, scale=1.0 / math.sqrt(Cn * 9))
    bias = rnd(N, seed=43, scale=0.5)
    xq, wq = q(x, dtype), q(w, dtype)
    wk = wq.permute(0, 2, 3, 1).contiguous().to(DEV, dtype)
    ref = F.conv2d(xq, wq, None, 1, 1)
    wide_x = torch.zeros(B, H, W, Cn + 16, dtype=dtype, device=DEV)
    wide_x[..., 8:8 + Cn] = nhwc(x, dtype)
    xv = wide_x[..., 8:8 + Cn]                                  # a channel slice: pixel stride Cn + 16
    try:
        _lib.set_option("igemm_cfg", 19)
        wide_y = torch.full((B, H, W, N + 24), 7.0, dtype=dtype, device=DEV)
        yv = wide_y[..., 16:16 + N]
        ssum, ssq = torch.zeros(N, device=DEV), torch.zeros(N, device=DEV)
        o.conv2d_fwd(xv, wk, yv, 3, 1, 1, stats=(ssum, ssq))
        close(to_nchw(yv), ref, dtype, "fwd into a slice")
        assert bool((wide_y[..., :16] == 7.0).all()) and bool((wide_y[..., 16 + N:] == 7.0).all())      # neighbours untouched
        close(ssum.cpu(), ref.sum((0, 2, 3)), dtype, "statistics", mult=4 * math.sqrt(B * H * W))
        close(ssq.cpu(), (ref * ref).sum((0, 2, 3)), dtype, "statistics sq", mult=8)
        y2 = torch.empty(B, H, W, N, dtype=dtype, device=DEV)
        o.conv2d_fwd(xv, wk, y2, 3, 1, 1, bias=bias.to(DEV), silu=True)
        close(to_nchw(y2), F.silu(ref + bias.view(1, -1, 1, 1)), dtype, "bias + SiLU", mult=2)
        dy = rnd(B, N, H, W, seed=44)
        ref_dx = torch.nn.grad.conv2d_input((B, Cn, H, W), wq, q(dy, dtype), 1, 1)
        dx = torch.zeros(B, H, W, Cn, dtype=dtype, device=DEV)
        wt = o.weight_transpose(wk)
        o.conv2d_dgrad(nhwc(dy, dtype), wt, dx, (B, H, W, N), 3, 1, 1)
        close(to_nchw(dx), ref_dx, dtype, "dgrad")
        o.conv2d_dgrad(nhwc(dy, dtype), wt, dx, (B, H, W, N), 3, 1, 1, accumulate=True)
        close(to_nchw(dx), 2 * ref_dx, dtype, "dgrad accumulate", mult=2)
    finally:
        _lib.set_option("igemm_cfg", -1)


@pytest.mark.parametrize("case", [
    # B, C, H, W, N, k, s  — every case is LEGAL for the 8-wave pipeline (M >= 256, K >= 128, N > 32): a forced illegal cfg would fall back silently
    (2, 96, 24, 20, 160, 3, 1),      # cfg 20: partial pixel tile (960 rows), partial channel tile (160 = 128 + 32), K = 864 = 13.5 stages
    (1, 40, 18, 30, 72, 3, 2),       # cfg 20: C = 40 < one stage (several taps per stage, the per-chunk tap walk), stride 2, N = 72
    (3, 512, 10, 10, 128, 1, 1),     # cfg 20: 1x1, 8 stages, 300 rows
    (1, 64, 16, 32, 136, 3, 1),      # cfg 20: exactly two pixel tiles, 9 stages (ring wraps three times), N = 136
    (2, 128, 20, 20, 64, 3, 1),      # cfg 21: 64-channel tile
    (1, 192, 40, 40, 48, 1, 1),      # cfg 21: N = 48 (partial 64-tile), K = 192 = 3 stages
    (2, 64, 13, 11, 128, 1, 1),      # cfg 20: K = 64: ONE stage... below the legal K (falls back): the dispatcher must refuse, not misbehave
    (1, 128, 16, 16, 256, 1, 1),     # cfg 20: K = 128 = two stages, exactly one pixel tile, two channel tiles
])
def test_eight_wave_pipeline_epilogues_and_tails(case):
    """igemm configurations 20 / 21 (igemm8.hip): the four epilogues it carries — BN statistics (training forward), none and accumulate
    (input gradients), bias + SiLU (the fused inference conv) — into channel slices of wider tensors (concat by pointer), with
    ragged pixel / channel / K tails, tap-major and channel-major K order."""
    from sy11 import _lib
    o = ops()
    dtype = torch.float16
    B, Cn, H, W, N, k, s = case
    p = k // 2
    x = rnd(B, Cn, H, W, seed=51)
    w = rnd(N, Cn, k, k, seed=52, scale=1.0 / math.sqrt(Cn * k * k))
    bias = rnd(N, seed=53, scale=0.5)
    xq, wq = q(x, dtype), q(w, dtype)
    OH, OW = o.conv_out_hw(H, W, k, s, p)
    wk = wq.permute(0, 2, 3, 1).contiguous().to(DEV, dtype)
    ref = F.conv2d(xq, wq, None, s, p)
    xv = nhwc(x, dtype, 16)                                     # a channel slice: pixel stride C + 16
    dy = rnd(B, N, OH, OW, seed=54)
    ref_dx = torch.nn.grad.conv2d_input((B, Cn, H, W), wq, q(dy, dtype), s, p)
    wt = o.weight_transpose(wk)
    korder0 = _lib.get_option("igemm_korder")
    try:
        # K order x (128-byte stages in 3 slots | 64-byte stages in 6 slots | the 256 x 256 tile where the layer has > 128 channels)
        # ... | the persistent form of cfg 20, kb64 = 8)
        for korder, kb64 in ((1, 0), (0, 0), (1, 2), (0, 2), (1, 4), (0, 4), (1, 8), (0, 8)):
            cfg = 25 if (kb64 == 8 and N > 64) else 24 if (kb64 == 4 and N > 128) else (20 if N > 64 else 21) + (kb64 & 2)
            _lib.set_option("igemm_korder", korder)
            _lib.set_option("igemm_cfg", cfg)
            wide_y = torch.full((B, OH, OW, N + 24), 7.0, dtype=dtype, device=DEV)
            yv = wide_y[..., 16:16 + N]
            ssum, ssq = torch.zeros(N, device=DEV), torch.zeros(N, device=DEV)
            o.conv2d_fwd(xv, wk, yv, k, s, p, stats=(ssum, ssq))
            close(to_nchw(yv), ref, dtype, f"fwd into a slice (korder {korder})")
            assert bool((wide_y[..., :16] == 7.0).all()) and bool((wide_y[..., 16 + N:] == 7.0).all())
            close(ssum.cpu(), ref.sum((0, 2, 3)), dtype, "statistics", mult=4 * math.sqrt(B * OH * OW))
            close(ssq.cpu(), (ref * ref).sum((0, 2, 3)), dtype, "statistics sq", mult=8)
            y2 = torch.empty(B, OH, OW, N, dtype=dtype, device=DEV)
            o.conv2d_fwd(xv, wk, y2, k, s, p, bias=bias.to(DEV), silu=True)
            close(to_nchw(y2), F.silu(ref + bias.view(1, -1, 1, 1)), dtype, "bias + SiLU", mult=2)
            # input gradient: the same kernel on dy (C and N swap roles: legal when C > 32 ... ) — plain, then accumulated
            _lib.set_option("igemm_cfg", 25 if (kb64 == 8 and Cn > 64) else 24 if (kb64 == 4 and Cn > 128) else (20 if Cn > 64 else 21) + (kb64 & 2))
            dx = torch.zeros(B, H, W, Cn, dtype=dtype, device=DEV)
            o.conv2d_dgrad(nhwc(dy, dtype), wt, dx, (B, OH, OW, N), k, s, p, accumulate=(s > 1))
            close(to_nchw(dx), ref_dx, dtype, f"dgrad (korder {korder})")
            o.conv2d_dgrad(nhwc(dy, dtype), wt, dx, (B, OH, OW, N), k, s, p, accumulate=True)
            close(to_nchw(dx), 2 * ref_dx, dtype, "dgrad accumulate", mult=2)
    finally:
        _lib.set_option("igemm_cfg", -1)
        _lib.set_option("igemm_korder", korder0)


@pytest.mark.parametrize("case", [(2, 64, 10, 80, 64), (1, 32, 8, 160, 16), (2, 128, 20, 20, 96), (1, 48, 12, 40, 72)])
def test_patch_filter_gradient_bf16_and_slices(case):
    """wgrad configurations 12-15 (wgrad3x3p_kernel) in bf16, on operands that are channel slices of wider tensors, with partial
    32 / 64 blocks (C = 48, N = 72, 96); accumulates into dW."""
    from sy11 import _lib
    o = ops()
    dtype = torch.bfloat16
    B, Cn, H, W, N = case
    x, dy = rnd(B, Cn, H, W, seed=51), rnd(B, N, H, W, seed=52)
    ref = torch.nn.grad.conv2d_weight(q(x, dtype), (N, Cn, 3, 3), q(dy, dtype), 1, 1)
    wide_x = torch.zeros(B, H, W, Cn + 8, dtype=dtype, device=DEV)
    wide_x[..., 8:] = nhwc(x, dtype)
    wide_dy = torch.zeros(B, H, W, N + 16, dtype=dtype, device=DEV)
    wide_dy[..., :N] = nhwc(dy, dtype)
    try:
        for cfg in range(12, 16):
            _lib.set_option("wgrad_cfg", cfg)
            dw = torch.zeros(N, 3, 3, Cn, dtype=torch.float32, device=DEV)
            o.conv2d_wgrad(wide_x[..., 8:], wide_dy[..., :N], dw, 3, 1, 1)
            close(dw.cpu().permute(0, 3, 1, 2), ref, torch.float32, f"wgrad cfg {cfg}", mult=200)      # bf16 operands: 8 mantissa bits
            o.conv2d_wgrad(wide_x[..., 8:], wide_dy[..., :N], dw, 3, 1, 1)
            close(dw.cpu().permute(0, 3, 1, 2), 2 * ref, torch.float32, f"wgrad cfg {cfg} accumulate", mult=400)
    finally:
        _lib.set_option("wgrad_cfg", -1)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("case", [(2, 256, 20, 20, 256, 3, 1), (1, 40, 23, 31, 136, 3, 2), (3, 384, 13, 17, 128, 1, 1), (2, 72, 16, 16, 200, 3, 1)])
def test_wide_tile_filter_gradient(case, dtype):
    """wgrad configurations 16-19 (wgrad16w_kernel, 128 filters x 256 columns, eight waves): whole and partial tiles in both
    directions (N = 136, 200; K = 360, 384, 648), stride 2, operands that are channel slices of wider tensors; accumulates."""
    from sy11 import _lib
    o = ops()
    B, Cn, H, W, N, k, s = case
    p = k // 2
    OH, OW = o.conv_out_hw(H, W, k, s, p)
    x, dy = rnd(B, Cn, H, W, seed=61), rnd(B, N, OH, OW, seed=62)
    ref = torch.nn.grad.conv2d_weight(q(x, dtype), (N, Cn, k, k), q(dy, dtype), s, p)
    wide_x = torch.zeros(B, H, W, Cn + 8, dtype=dtype, device=DEV)
    wide_x[..., 8:] = nhwc(x, dtype)
    wide_dy = torch.zeros(B, OH, OW, N + 16, dtype=dtype, device=DEV)
    wide_dy[..., :N] = nhwc(dy, dtype)
    mult = 200 if dtype == torch.bfloat16 else 8
    try:
        for cfg in range(16, 20):
            _lib.set_option("wgrad_cfg", cfg)
            dw = torch.zeros(N, k, k, Cn, dtype=torch.float32, device=DEV)
            o.conv2d_wgrad(wide_x[..., 8:], wide_dy[..., :N], dw, k, s, p)
            close(dw.cpu().permute(0, 3, 1, 2), ref, torch.float32, f"wgrad cfg {cfg}", mult=mult)
            o.conv2d_wgrad(wide_x[..., 8:], wide_dy[..., :N], dw, k, s, p)
            close(dw.cpu().permute(0, 3, 1, 2), 2 * ref, torch.float32, f"wgrad cfg {cfg} accumulate", mult=2 * mult)
    finally:
        _lib.set_option("wgrad_cfg", -1)


@pytest.mark.parametrize("case", [(2, 64, 32, 32, 96), (1, 32, 80, 80, 64), (1, 32, 31, 39, 64), (2, 256, 40, 40, 128), (3, 128, 32, 48, 32)])
def test_stride2_input_gradient_all_parities_in_one_pass(case):
    """halo_dgrad_s2_kernel (conv3x3.hip): dx of a 3x3 / stride 2 / pad 1 conv, four parity classes from one dy patch, whole
    pixel rows written — against the fp32 CPU gradient, plain and accumulating, and bit-for-bit-close to the four-launch igemm path."""
    from sy11 import _lib
    o = ops()
    dtype = torch.float16
    B, Cn, H, W, N = case
    k, s, p = 3, 2, 1
    OH, OW = o.conv_out_hw(H, W, k, s, p)
    w = rnd(N, Cn, k, k, seed=22, scale=1.0 / math.sqrt(Cn * 9))
    dy = rnd(B, N, OH, OW, seed=23)
    wq, dyq = q(w, dtype), q(dy, dtype)
    dyv = nhwc(dy, dtype, 16)
    wk = wq.permute(0, 2, 3, 1).contiguous().to(DEV, dtype)
    wt = o.weight_transpose(wk)
    ref_dx = torch.nn.grad.conv2d_input((B, Cn, H, W), wq, dyq, s, p)
    outs = []
    try:
        for flag in (2, 0):
            _lib.set_option("dgrad_s2_halo", flag)
            dx = torch.zeros(B, H, W, Cn + 8, dtype=dtype, device=DEV)[..., :Cn]          # ld > C
            o.conv2d_dgrad(dyv, wt, dx, (B, OH, OW, N), k, s, p)
            close(to_nchw(dx), ref_dx, dtype, f"dgrad s2 (halo {flag})")
            o.conv2d_dgrad(dyv, wt, dx, (B, OH, OW, N), k, s, p, accumulate=True)
            close(to_nchw(dx), 2 * ref_dx, dtype, f"dgrad s2 accumulate (halo {flag})", mult=2)
            outs.append(to_nchw(dx))
    finally:
        _lib.set_option("dgrad_s2_halo", 1)
    assert (outs[0] - outs[1]).abs().max().item() <= 4e-3 * ref_dx.abs().max().item()


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("Cn,H,W", [(64, 8, 8), (24, 7, 5), (5, 6, 6), (256, 20, 20), (128, 13, 27), (512, 3, 12)])
def test_depthwise_conv(Cn, H, W, dtype):
    o = ops()
    B, k, p = 2, 3, 1
    x, w, dy = rnd(B, Cn, H, W, seed=1), rnd(Cn, 1, k, k, seed=2, scale=0.3), rnd(B, Cn, H, W, seed=3)
    xq, wq, dyq = q(x, dtype), q(w, dtype), q(dy, dtype)
    xv, dyv = nhwc(x, dtype), nhwc(dy, dtype)
    wk = wq.permute(0, 2, 3, 1).contiguous().to(DEV, dtype)
    y = torch.empty(B, H, W, Cn, dtype=dtype, device=DEV)
    ssum, ssq = torch.zeros(Cn, device=DEV), torch.zeros(Cn, device=DEV)
    o.conv2d_fwd(xv, wk, y, k, 1, p, groups=Cn, stats=(ssum, ssq))
    ref = F.conv2d(xq, wq, None, 1, p, 1, Cn)
    close(to_nchw(y), ref, dtype, "dw fwd")
    close(ssum.cpu(), ref.sum((0, 2, 3)), dtype, "dw stats", mult=20)
    dx = torch.empty(B, H, W, Cn, dtype=dtype, device=DEV)
    o.conv2d_dgrad(dyv, wk, dx, (B, H, W, Cn), k, 1, p, groups=Cn)
    close(to_nchw(dx), torch.nn.grad.conv2d_input((B, Cn, H, W), wq, dyq, 1, p, 1, Cn), dtype, "dw dgrad")
    dw = torch.zeros(Cn, k, k, 1, dtype=torch.float32, device=DEV)
    o.conv2d_wgrad(xv, dyv, dw, k, 1, p, groups=Cn)
    close(dw.cpu().permute(0, 3, 1, 2), torch.nn.grad.conv2d_weight(xq, (Cn, 1, k, k), dyq, 1, p, 1, Cn), torch.float32,
          "dw wgrad", mult=8)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16])
@pytest.mark.parametrize("N,B,H,W,s,p", [(32, 3, 37, 264, 2, 1), (64, 2, 26, 300, 2, 1), (32, 2, 19, 150, 1, 1), (64, 1, 9, 131, 1, 0),
                                         (32, 2, 64, 512, 2, 1), (32, 1, 11, 13, 2, 0)])
def test_stem_conv_lds_tiles(N, B, H, W, s, p, dtype):
    """The LDS-tiled stem kernels (direct.hip stem_fwd_tile / stem_wgrad_tile: 4 x 64 output pixels per workgroup) on maps several
    tiles wide and high, ragged last tiles, widths that are not multiples of 4 (scalar fill), both strides, with and without
    padding — forward, the BatchNorm statistics and the filter gradient against fp32 autograd on the rounded operands."""
    o = ops()
    x = (rnd(B, 3, H, W, seed=31) + 1) / 2
    w = rnd(N, 3, 3, 3, seed=32, scale=0.3)
    wq, xq = q(w, dtype), q(x, dtype)
    OH, OW = o.conv_out_hw(H, W, 3, s, p)
    wk = wq.permute(0, 2, 3, 1).contiguous().to(DEV, dtype)
    y = torch.full((B, OH, OW, N), float("nan"), dtype=dtype, device=DEV)
    ssum, ssq = torch.zeros(N, device=DEV), torch.zeros(N, device=DEV)
    o.stem_conv_fwd(x.to(DEV), wk, y, s, p, stats=(ssum, ssq))
    ref = F.conv2d(xq, wq, None, s, p)
    close(to_nchw(y), ref, dtype, "stem fwd")
    close(ssum.cpu(), ref.sum((0, 2, 3)), dtype, "stem sum", mult=4)
    close(ssq.cpu(), (ref * ref).sum((0, 2, 3)), dtype, "stem sumsq", mult=4)
    dy = rnd(B, N, OH, OW, seed=33)
    dw = torch.zeros(N, 3, 3, 3, dtype=torch.float32, device=DEV)
    o.stem_conv_wgrad(x.to(DEV), nhwc(dy, dtype), dw, s, p)
    close(dw.cpu().permute(0, 3, 1, 2), torch.nn.grad.conv2d_weight(xq, (N, 3, 3, 3), q(dy, dtype), s, p), torch.float32,
          "stem wgrad", mult=8)
    dw2 = torch.zeros_like(dw)                                 # accumulates, and twice the same
    o.stem_conv_wgrad(x.to(DEV), nhwc(dy, dtype), dw2, s, p)
    o.stem_conv_wgrad(x.to(DEV), nhwc(dy, dtype), dw2, s, p)
    assert torch.allclose(dw2, 2 * dw, rtol=1e-4, atol=1e-4 * float(dw.abs().max()))


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("N,H,W", [(16, 16, 16), (32, 21, 18), (32, 70, 66), (64, 40, 36)])
def test_stem_conv(N, H, W, dtype):
    o = ops()
    B = 2
    x = (rnd(B, 3, H, W, seed=1) + 1) / 2
    w = rnd(N, 3, 3, 3, seed=2, scale=0.3)
    wq = q(w, dtype)
    OH, OW = o.conv_out_hw(H, W, 3, 2, 1)
    wk = wq.permute(0, 2, 3, 1).contiguous().to(DEV, dtype)
    y = torch.empty(B, OH, OW, N, dtype=dtype, device=DEV)
    ssum, ssq = torch.zeros(N, device=DEV), torch.zeros(N, device=DEV)
    o.stem_conv_fwd(x.to(DEV), wk, y, 2, 1, stats=(ssum, ssq))
    # N = 32 / 64 in 16 bit runs on the MFMA, which rounds the image to the compute dtype exactly as autocast does in the
    # reference; the other widths take the f32 VALU kernel that multiplies the f32 image directly
    xq = q(x, dtype) if N in (32, 64) else x
    ref = F.conv2d(xq, wq, None, 2, 1)
    close(to_nchw(y), ref, dtype, "stem fwd")
    close(ssq.cpu(), (ref * ref).sum((0, 2, 3)), dtype, "stem stats", mult=4)
    dy = rnd(B, N, OH, OW, seed=3)
    dw = torch.zeros(N, 3, 3, 3, dtype=torch.float32, device=DEV)
    o.stem_conv_wgrad(x.to(DEV), nhwc(dy, dtype), dw, 2, 1)
    close(dw.cpu().permute(0, 3, 1, 2), torch.nn.grad.conv2d_weight(xq, (N, 3, 3, 3), q(dy, dtype), 2, 1), torch.float32,
          "stem wgrad", mult=8)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("Cn,silu", [(64, True), (48, False), (5, True), (1024, True)])
def test_batchnorm_silu_fwd_bwd(Cn, silu, dtype):
    o = ops()
    B, H, W = 2, 5, 7
    M = B * H * W
    y = rnd(B, Cn, H, W, seed=1, scale=2.0)
    yq = q(y, dtype)
    gamma, beta = 1 + 0.3 * rnd(Cn, seed=2), 0.2 * rnd(Cn, seed=3)
    res = rnd(B, Cn, H, W, seed=5)
    dz = rnd(B, Cn, H, W, seed=4)
    rm, rv = 0.1 * rnd(Cn, seed=6), 1 + 0.2 * rnd(Cn, seed=7).abs()
    # reference (autograd, fp32)
    yr = yq.clone().requires_grad_(True)
    g_, b_ = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    u = F.batch_norm(yr, rm_ref, rv_ref, g_, b_, True, 0.03, 1e-3)
    z = (F.silu(u) if silu else u) + q(res, dtype)
    (z * q(dz, dtype)).sum().backward()
    # device
    f = lambda t: t.to(DEV)
    yv, dzv, resv = nhwc(y, dtype, 8), nhwc(dz, dtype), nhwc(res, dtype)
    ssum = yv.float().sum((0, 1, 2))
    ssq = (yv.float() ** 2).sum((0, 1, 2))
    mean, rstd, scale, shift = (torch.empty(Cn, device=DEV) for _ in range(4))
    rmd, rvd = f(rm), f(rv)
    o.bn_finalize(M, ssum, ssq, f(gamma), f(beta), 1e-3, 0.03, rmd, rvd, mean, rstd, scale, shift)
    close(rmd.cpu(), rm_ref, torch.float32, "running_mean", mult=10)
    close(rvd.cpu(), rv_ref, torch.float32, "running_var", mult=10)
    zv = torch.empty(B, H, W, Cn, dtype=dtype, device=DEV)
    o.bn_act_fwd(yv, scale, shift, zv, silu=silu, res=resv)
    close(to_nchw(zv), z.detach(), dtype, "bn fwd", mult=2)
    sg, sgx = torch.zeros(Cn, device=DEV), torch.zeros(Cn, device=DEV)
    o.bn_act_bwd_reduce(yv, dzv, mean, rstd, scale, shift, silu, sg, sgx)
    dyv = torch.empty(B, H, W, Cn, dtype=dtype, device=DEV)
    dg, db = torch.zeros(Cn, device=DEV), torch.zeros(Cn, device=DEV)
    o.bn_act_bwd_apply(yv, dzv, mean, rstd, scale, shift, f(gamma), silu, sg, sgx, dyv, dg, db)
    close(db.cpu(), b_.grad, dtype, "dbeta", mult=10)
    close(dg.cpu(), g_.grad, dtype, "dgamma", mult=10)
    close(to_nchw(dyv), yr.grad, dtype, "bn dy", mult=6)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("accumulate", [False, True])
@pytest.mark.parametrize("Cn,B,H,W", [(136, 3, 37, 41), (24, 5, 61, 67), (64, 16, 80, 80), (2056, 2, 9, 11)])
def test_batchnorm_backward_apply_writes_the_residual_gradient(Cn, B, H, W, accumulate, dtype):
    """sy11_bn_act_bwd_apply_res: dy as without the residual output, and res_grad (= | +=) dz bit-identical to what the separate
    sy11_copy2d launch produced (f32 add, one rounding), into a channel slice of a wider buffer."""
    o = ops()
    M = B * H * W
    y, dz = rnd(B, Cn, H, W, seed=61, scale=1.5), rnd(B, Cn, H, W, seed=62)
    gamma, beta = 1 + 0.3 * rnd(Cn, seed=63), 0.2 * rnd(Cn, seed=64)
    f = lambda t: t.to(DEV)
    yv, dzv = nhwc(y, dtype), nhwc(dz, dtype)
    ssum, ssq = yv.float().sum((0, 1, 2)), (yv.float() ** 2).sum((0, 1, 2))
    mean, rstd, scale, shift = (torch.empty(Cn, device=DEV) for _ in range(4))
    o.bn_finalize(M, ssum, ssq, f(gamma), f(beta), 1e-3, 0.03, None, None, mean, rstd, scale, shift)
    sg = torch.zeros(2, 8, Cn, device=DEV)
    o.bn_act_bwd_reduce(yv, dzv, mean, rstd, scale, shift, True, sg[0], sg[1])
    old = nhwc(rnd(B, Cn, H, W, seed=65), dtype)
    wide = torch.full((B, H, W, Cn + 8), 3.0, dtype=dtype, device=DEV)
    wide[..., :Cn] = old
    ref_dy = torch.empty(B, H, W, Cn, dtype=dtype, device=DEV)
    o.bn_act_bwd_apply(yv, dzv, mean, rstd, scale, shift, f(gamma), True, sg[0], sg[1], ref_dy, None, None)
    ref_res = old.clone()
    o.copy2d(dzv, ref_res, accumulate=accumulate)
    dy = torch.empty_like(ref_dy)
    o.bn_act_bwd_apply(yv, dzv, mean, rstd, scale, shift, f(gamma), True, sg[0], sg[1], dy, None, None,
                       res_grad=wide[..., :Cn], res_accumulate=accumulate)
    assert torch.equal(dy, ref_dy)
    assert torch.equal(wide[..., :Cn], ref_res)
    assert bool((wide[..., Cn:] == 3.0).all())


@pytest.mark.parametrize("row_map", [0, 1])
@pytest.mark.parametrize("Cn,B,H,W,slots", [(136, 3, 37, 41, 8), (2056, 2, 9, 11, 1), (24, 5, 61, 67, 8), (64, 16, 80, 80, 8)])
def test_batchnorm_row_walk(Cn, B, H, W, slots, row_map):
    """The two row walks of the BatchNorm passes (elementwise.hip RowWalk: contiguous chunks / the strided r01 grid) on maps with
    several workgroups, ragged last chunks, two channel blocks (C = 2056 > 256 vectors) and slotted sums, against fp32 autograd."""
    from sy11 import _lib
    o = ops()
    dtype = torch.float16
    M = B * H * W
    y = rnd(B, Cn, H, W, seed=11, scale=1.5)
    yq = q(y, dtype)
    gamma, beta = 1 + 0.3 * rnd(Cn, seed=12), 0.2 * rnd(Cn, seed=13)
    dz = rnd(B, Cn, H, W, seed=14)
    yr = yq.clone().requires_grad_(True)
    g_, b_ = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    u = F.batch_norm(yr, None, None, g_, b_, True, 0.03, 1e-3)
    z = F.silu(u)
    (z * q(dz, dtype)).sum().backward()
    f = lambda t: t.to(DEV)
    yv, dzv = nhwc(y, dtype), nhwc(dz, dtype)
    ssum = yv.float().sum((0, 1, 2))
    ssq = (yv.float() ** 2).sum((0, 1, 2))
    mean, rstd, scale, shift = (torch.empty(Cn, device=DEV) for _ in range(4))
    o.bn_finalize(M, ssum, ssq, f(gamma), f(beta), 1e-3, 0.03, None, None, mean, rstd, scale, shift)
    try:
        _lib.set_option("row_map", row_map)
        zv = torch.empty(B, H, W, Cn, dtype=dtype, device=DEV)
        o.bn_act_fwd(yv, scale, shift, zv, silu=True)
        close(to_nchw(zv), z.detach(), dtype, "bn fwd", mult=2)
        sg = torch.zeros(2, slots, Cn, device=DEV)
        o.bn_act_bwd_reduce(yv, dzv, mean, rstd, scale, shift, True, sg[0], sg[1])
        dyv = torch.empty(B, H, W, Cn, dtype=dtype, device=DEV)
        dg, db = torch.zeros(Cn, device=DEV), torch.zeros(Cn, device=DEV)
        o.bn_act_bwd_apply(yv, dzv, mean, rstd, scale, shift, f(gamma), True, sg[0], sg[1], dyv, dg, db)
    finally:
        _lib.set_option("row_map", 1)
    close(db.cpu(), b_.grad, dtype, "dbeta", mult=10)
    close(dg.cpu(), g_.grad, dtype, "dgamma", mult=10)
    close(to_nchw(dyv), yr.grad, dtype, "bn dy", mult=6)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_copy_upsample_maxpool(dtype):
    o = ops()
    B, Cn, H, W = 2, 40, 6, 9
    x = rnd(B, Cn, H, W, seed=1)
    xv = nhwc(x, dtype, 8)
    # copy into a slice, then accumulate
    buf = torch.zeros(B, H, W, Cn + 24, dtype=dtype, device=DEV)
    o.copy2d(xv, buf[..., 16:16 + Cn])
    assert torch.equal(buf[..., 16:16 + Cn], xv) and buf[..., :16].abs().max() == 0
    o.copy2d(xv, buf[..., 16:16 + Cn], accumulate=True)
    close(to_nchw(buf[..., 16:16 + Cn]), 2 * q(x, dtype), dtype, "copy accumulate")
    # upsample fwd / bwd
    up = torch.empty(B, 2 * H, 2 * W, Cn, dtype=dtype, device=DEV)
    o.upsample2x_fwd(xv, up)
    assert torch.equal(to_nchw(up), F.interpolate(q(x, dtype), scale_factor=2.0, mode="nearest"))
    g = rnd(B, Cn, 2 * H, 2 * W, seed=2)
    dx = torch.empty(B, H, W, Cn, dtype=dtype, device=DEV)
    o.upsample2x_bwd(nhwc(g, dtype), dx)
    close(to_nchw(dx), F.avg_pool2d(q(g, dtype), 2) * 4, dtype, "upsample bwd")
    # max-pool 5x5 s1 p2 with plateaus (ties): quantise the input so equal values are common
    xt = (rnd(B, Cn, H, W, seed=3) * 3).round() / 3
    xtv = nhwc(xt, dtype)
    yv = torch.empty(B, H, W, Cn, dtype=dtype, device=DEV)
    idx = torch.empty(B, H, W, Cn, dtype=torch.uint8, device=DEV)
    o.maxpool5_fwd(xtv, yv, idx)
    xr = q(xt, dtype).requires_grad_(True)
    yr = F.max_pool2d(xr, 5, 1, 2)
    assert torch.equal(to_nchw(yv), yr.detach())
    gp = rnd(B, Cn, H, W, seed=4)
    yr.backward(q(gp, dtype))
    dxp = torch.empty(B, H, W, Cn, dtype=dtype, device=DEV)
    o.maxpool5_bwd(nhwc(gp, dtype), idx, dxp)
    close(to_nchw(dxp), xr.grad, dtype, "maxpool bwd (tie routing)", mult=4)


@pytest.mark.parametrize("dtype", [torch.float16, torch.bfloat16, torch.float32])
@pytest.mark.parametrize("B,Cn,H,W", [(3, 64, 20, 20), (2, 24, 7, 3), (1, 8, 1, 1), (2, 16, 5, 23), (1, 40, 11, 4), (2, 8, 2, 9)])
def test_maxpool5_values_and_argmax_against_aten(B, Cn, H, W, dtype):
    """5 x 5 stride-1 max-pool: values bit-exact and the stored window position = ATen's argmax (first maximum in row-major window
    order) on inputs full of ties, zeros of both signs and negative plateaus — the separable 16-bit kernel and the 25-way f32 one."""
    o = ops()
    xt = (rnd(B, Cn, H, W, seed=71) * 2).round() / 2                  # plateaus
    xt[:, : Cn // 2] = -xt[:, : Cn // 2].abs()                         # all-negative channels (a zero pad would win there)
    xt[0, -1] = 0.0
    xt[0, -1, ::2] = -0.0
    xv = nhwc(xt, dtype)
    yv = torch.empty(B, H, W, Cn, dtype=dtype, device=DEV)
    idx = torch.empty(B, H, W, Cn, dtype=torch.uint8, device=DEV)
    o.maxpool5_fwd(xv, yv, idx)
    ref, ref_i = F.max_pool2d(q(xt, dtype), 5, 1, 2, return_indices=True)
    assert torch.equal(to_nchw(yv).float(), ref)
    pos = idx.permute(0, 3, 1, 2).cpu().long()                          # (B, C, H, W) window positions 0..24
    hh = torch.arange(H).view(1, 1, H, 1) + pos // 5 - 2
    ww = torch.arange(W).view(1, 1, 1, W) + pos % 5 - 2
    assert bool(((hh >= 0) & (hh < H) & (ww >= 0) & (ww < W)).all())
    got = hh * W + ww
    # ATen keeps the FIRST maximum; +0 / -0 compare equal there: compare indices where the maximum is unique in value AND sign-
    # insensitive, i.e. everywhere except windows whose maximum is a zero of mixed sign
    flat = q(xt, dtype).reshape(B, Cn, H * W)
    same_val = torch.gather(flat, 2, got.reshape(B, Cn, -1)) == torch.gather(flat, 2, ref_i.reshape(B, Cn, -1))
    assert bool(same_val.all())
    nonzero_max = (ref != 0)
    assert torch.equal(got[nonzero_max], ref_i[nonzero_max])
    # backward: routes dy to the argmax, plain and accumulating, against autograd of the same pooling
    xr = q(xt, dtype).requires_grad_(True)
    gp = rnd(B, Cn, H, W, seed=72)
    F.max_pool2d(xr, 5, 1, 2).backward(q(gp, dtype))
    dxv = torch.empty(B, H, W, Cn, dtype=dtype, device=DEV)
    o.maxpool5_bwd(nhwc(gp, dtype), idx, dxv)
    routed = torch.zeros(B, Cn, H * W).scatter_add_(2, got.reshape(B, Cn, -1), q(gp, dtype).reshape(B, Cn, -1)).reshape(B, Cn, H, W)
    close(to_nchw(dxv), routed, dtype, "maxpool bwd vs its own argmax", mult=4)
    ok = nonzero_max.reshape(B, Cn, -1).all(2)                          # channels without signed-zero ties: ATen routes the same way
    close(to_nchw(dxv)[ok], xr.grad[ok], dtype, "maxpool bwd vs autograd", mult=4)
    old = nhwc(rnd(B, Cn, H, W, seed=73), dtype)
    acc = old.clone()
    o.maxpool5_bwd(nhwc(gp, dtype), idx, acc, accumulate=True)
    close(to_nchw(acc), to_nchw(old) + routed, dtype, "maxpool bwd accumulate", mult=6)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("B,H,W,heads", [(2, 6, 5, 2), (1, 20, 20, 4), (2, 3, 3, 1), (3, 10, 10, 2), (2, 8, 4, 1), (1, 40, 40, 2), (2, 37, 37, 1), (1, 50, 50, 1)])
def test_attention_fwd_bwd(B, H, W, heads, dtype):
    """C2PSA attention core against autograd.  Past ~1 200 tokens the f32 kernels (and past 2 400 the 16-bit backward) keep their
    score rows in P / the dS workspace instead of LDS (r04): 40 x 40 and 50 x 50 maps, and 37 x 37 with a partial last query tile."""
    o = ops()
    kd, hd = 32, 64
    N = H * W
    Cq = heads * (2 * kd + hd)
    qkv = rnd(B, Cq, H, W, seed=1)
    qq = q(qkv, dtype).requires_grad_(True)
    qh, kh, vh = qq.view(B, heads, 2 * kd + hd, N).split([kd, kd, hd], 2)
    attn = ((qh.transpose(-2, -1) @ kh) * kd ** -0.5).softmax(-1)
    oref = (vh @ attn.transpose(-2, -1)).view(B, heads * hd, H, W)
    do = rnd(B, heads * hd, H, W, seed=2)
    (oref * q(do, dtype)).sum().backward()
    qv = nhwc(qkv, dtype)
    ov = torch.empty(B, H, W, heads * hd, dtype=dtype, device=DEV)
    p = torch.empty(B, heads, N, N, dtype=torch.float32, device=DEV)
    o.attention_fwd(qv, heads, kd, hd, ov, p)
    close(p.cpu(), attn.detach(), dtype, "attention P", mult=4)
    close(to_nchw(ov), oref.detach(), dtype, "attention o", mult=4)
    dq = torch.empty(B, H, W, Cq, dtype=dtype, device=DEV)
    ws = torch.empty(B, heads, N, N, dtype=torch.float32, device=DEV)
    o.attention_bwd(qv, heads, kd, hd, p, nhwc(do, dtype), dq, ws)
    close(to_nchw(dq), qq.grad, dtype, "attention dqkv", mult=8)
    dq2 = torch.empty_like(dq)                      # with the forward output: row sums as dO . o (the MFMA path reads P once)
    o.attention_bwd(qv, heads, kd, hd, p, nhwc(do, dtype), dq2, ws, o=ov)
    close(to_nchw(dq2), qq.grad, dtype, "attention dqkv (row sums from o)", mult=8)


def test_detect_decode_matches_oracle():
    from oracle import yolo11_ref as R
    o = ops()
    nc, B = 7, 2
    maps = [rnd(B, 64 + nc, h, w, seed=i, scale=3.0) for i, (h, w) in enumerate([(8, 6), (4, 3), (2, 2)])]
    ref = R.detect_decode(maps, (8.0, 16.0, 32.0), nc)
    dev = [m.permute(0, 2, 3, 1).contiguous().to(DEV) for m in maps]
    got = o.detect_decode(dev, (8.0, 16.0, 32.0), nc)
    close(got.cpu(), ref, torch.float32, "detect decode", mult=4)


@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 700, 3000])
def test_nms_bit_exact_vs_oracle(n):
    from oracle import nms_ref
    o = ops()
    rng = np.random.default_rng(n)
    xy = rng.uniform(0, 300, (n, 2)).astype(np.float32)
    wh = rng.uniform(4, 90, (n, 2)).astype(np.float32)
    boxes = np.concatenate([xy, xy + wh], 1)
    scores = rng.uniform(0, 1, n).astype(np.float32)
    if n > 10:
        scores[5] = scores[3]            # a tie: lower original index must win
        boxes[5] = boxes[3]
    ref_keep = nms_ref.nms_core(boxes, scores, 0.6)
    order = torch.sort(torch.from_numpy(scores), descending=True, stable=True).indices
    keep = o.nms_sorted(torch.from_numpy(boxes)[order].to(DEV), 0.6)
    got = order[keep.cpu()].numpy()
    assert np.array_equal(got, ref_keep)


def test_stft_logmel_matches_oracle():
    from oracle import stft_ref as S
    o = ops()
    iq = S.synthetic_iq(2, seed=1)
    ref_db = S.logmel_db(iq)                       # (B, mel, frames)
    start, wts = S.mel_table()
    win = torch.hann_window(S.N_FFT, periodic=True)
    db, mm = o.stft_logmel(iq.to(DEV), win.to(DEV), torch.from_numpy(start).to(DEV), torch.from_numpy(wts).to(DEV),
                           S.N_FFT, S.HOP, S.N_FRAMES, S.N_MEL)
    got = db.cpu().transpose(1, 2)
    # dB domain: f32 FFT of 1024 points -> ~1e-6 relative power error -> < 1e-3 dB except in deep nulls
    err = (got - ref_db).abs()
    assert err.max().item() < 5e-2 and err.mean().item() < 1e-4, (err.max().item(), err.mean().item())
    assert torch.allclose(mm[:, 0].cpu(), got.amin((1, 2))) and torch.allclose(mm[:, 1].cpu(), got.amax((1, 2)))
    img = o.stft_normalize(db, mm).cpu()
    ref_img = S.spectrogram_image(iq)
    assert img.shape == (2, 3, 640, 640)
    assert (img - ref_img).abs().max().item() < 2e-3
    assert img.min().item() == 0.0 and abs(img.max().item() - 1.0) < 1e-6


@pytest.mark.parametrize("case", ["dense", "grouped", "depthwise", "stem"])
def test_conv_fwd_bn_tail_matches_separate_finalize(case):
    """sy11_conv2d_fwd_bn / sy11_stem_conv_fwd_bn (statistics finalised by the last workgroup) == conv + sy11_bn_finalize."""
    o = ops()
    dtype = torch.float16
    B, H, W = 3, 20, 20
    Cn, N, k, s, p, g = {"dense": (32, 64, 3, 1, 1, 1), "grouped": (64, 32, 3, 2, 1, 8), "depthwise": (64, 64, 3, 1, 1, 64),
                         "stem": (3, 32, 3, 2, 1, 1)}[case]
    OH, OW = o.conv_out_hw(H, W, k, s, p)
    torch.manual_seed(0)
    w = (torch.randn(N, k, k, Cn // g, device=DEV) * 0.2).to(dtype)
    gamma, beta = torch.rand(N, device=DEV) + 0.5, torch.randn(N, device=DEV) * 0.1
    outs = []
    for fused in (False, True):
        rm, rv = torch.zeros(N, device=DEV), torch.ones(N, device=DEV)
        st = torch.zeros(2, 32, N, device=DEV)
        v = torch.empty(4, N, device=DEV)
        y = torch.empty(B, OH, OW, N, dtype=dtype, device=DEV)
        if case == "stem":
            x = torch.rand(B, 3, H, W, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
        else:
            x = torch.randn(B, H, W, Cn, device=DEV, generator=torch.Generator(device=DEV).manual_seed(1)).to(dtype)
        tail = (B * OH * OW, gamma, beta, 1e-3, 0.03, rm, rv, v[0], v[1], v[2], v[3], torch.zeros(max(g, 1), device=DEV))
        if fused:
            if case == "stem":
                o.stem_conv_fwd_bn(x, w, y, s, p, (st[0], st[1]), tail)
            else:
                o.conv2d_fwd_bn(x, w, y, k, s, p, 1, g, (st[0], st[1]), tail)
            assert float(tail[-1].abs().sum()) == 0.0                 # tickets returned at zero
        else:
            if case == "stem":
                o.stem_conv_fwd(x, w, y, s, p, stats=(st[0], st[1]))
            else:
                o.conv2d_fwd(x, w, y, k, s, p, 1, g, stats=(st[0], st[1]))
            o.bn_finalize(B * OH * OW, st[0], st[1], gamma, beta, 1e-3, 0.03, rm, rv, v[0], v[1], v[2], v[3])
        outs.append((y.float().cpu(), v.cpu(), rm.cpu(), rv.cpu()))
    assert torch.equal(outs[0][0], outs[1][0])
    for a, b_ in zip(outs[0][1:], outs[1][1:]):
        assert torch.allclose(a, b_, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("n_fft,hop,n_frames,n_mel", [
    (1024, 256, 37, 600),      # wave-per-frame kernel: frames not a multiple of the run (2) or of four waves x run, mel bank not a multiple of 64
    (1024, 256, 5, 640),       # the same with one frame per wave
    (1024, 256, 333, 512),     # runs of 5 frames, a partial last workgroup
    (1024, 128, 300, 640),     # hop != 256: the workgroup-per-frame kernel, four frames per workgroup
    (1024, 512, 9, 640),       # ... one frame per workgroup
    (256, 64, 50, 160),        # other FFT sizes: the generic radix-2 kernel
    (2048, 512, 12, 1280),
])
def test_stft_logmel_other_shapes_match_torch_stft(n_fft, hop, n_frames, n_mel):
    """Every dispatch branch of sy11_stft_logmel (stft.hip) on ragged shapes against torch.stft + the oracle's filter bank on the CPU:
    dB values, the per-image min / max the kernel reduces on the way, and the normalised image."""
    from oracle import stft_ref as S
    o = ops()
    B = 3
    L = n_fft + (n_frames - 1) * hop
    g = torch.Generator().manual_seed(n_fft + hop + n_frames)
    iq = torch.complex(torch.randn(B, L, generator=g), torch.randn(B, L, generator=g)) * 0.3
    t = torch.arange(L, dtype=torch.float32)
    iq = (iq + torch.exp(2j * math.pi * (0.11 * t + 1e-6 * t * t)).to(torch.complex64)).to(torch.complex64)      # a chirp above the noise
    win = torch.hann_window(n_fft, periodic=True)
    X = torch.stft(iq, n_fft, hop_length=hop, win_length=n_fft, window=win, center=False, onesided=False, return_complex=True)
    P = torch.fft.fftshift(X, dim=1).abs() ** 2
    ref = 10.0 * torch.log10(torch.einsum("jk,bkt->bjt", torch.from_numpy(S.mel_matrix(n_mel, n_fft)), P) + S.LOG_EPS)     # (B, mel, frames)
    start, wts = S.mel_table(n_mel, n_fft)
    db, mm = o.stft_logmel(iq.to(DEV), win.to(DEV), torch.from_numpy(start).to(DEV), torch.from_numpy(wts).to(DEV), n_fft, hop, n_frames, n_mel)
    got = db.cpu().transpose(1, 2)
    err = (got - ref).abs()
    assert err.max().item() < 5e-2 and err.mean().item() < 2e-4, (err.max().item(), err.mean().item())
    assert torch.allclose(mm[:, 0].cpu(), got.amin((1, 2))) and torch.allclose(mm[:, 1].cpu(), got.amax((1, 2)))
    img = o.stft_normalize(db, mm).cpu()
    lo, hi = ref.amin((1, 2), keepdim=True), ref.amax((1, 2), keepdim=True)
    ref_img = ((ref - lo) / (hi - lo).clamp(min=1e-12)).unsqueeze(1).expand(-1, 3, -1, -1)
    assert img.shape == (B, 3, n_mel, n_frames)
    assert (img - ref_img).abs().max().item() < 2e-3
