"""Convolution modules with the reference's names, constructor signatures and state_dict keys
(ultralytics/nn/modules/conv.py:56-83 autopad/Conv, :687-692 DWConv, :1810-1821 Concat), executed by libsy11.

``self.conv`` / ``self.bn`` are ordinary ``nn.Conv2d`` / ``nn.BatchNorm2d`` objects used as PARAMETER HOLDERS
(identical pickling / checkpoint layout); their ``forward`` is never called.  The filter is kept in
channels_last memory so the kernels read [Cout][KH][KW][Cin] in place.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from ... import ops
from ...engine import Act, Ctx, run_module

import os

_BN_TAIL = os.environ.get("SY11_BN_TAIL", "0") != "0"
_BN_STAT_SLOTS = max(1, int(os.environ.get("SY11_BN_STAT_SLOTS", "32")))     # statistic rows per layer; r04 sweep: 4 ... 32 rows are equal within noise, 1 row costs 0.65 ms per step (profiles/r04/bn_fused_finalize_experiment.txt)

__all__ = ("Conv", "DWConv", "DDWConv", "Concat", "WeightedSpatialAttention", "GCT", "Fusion", "autopad")


def autopad(k, p=None, d=1):
    """'same' padding (conv.py:56-62)."""
    if d > 1:
        k = d * (k - 1) + 1 if isinstance(k, int) else [d * (x - 1) + 1 for x in k]
    if p is None:
        p = k // 2 if isinstance(k, int) else [x // 2 for x in k]
    return p


def _int(v):
    return v[0] if isinstance(v, (tuple, list)) else v


def working_filter(holder: nn.Module, w: torch.Tensor, dtype: torch.dtype, force: bool = False) -> torch.Tensor:
    """[O][KH][KW][I] tensor of ``dtype`` for parameter ``w``; cached until the parameter is modified in place
    (``force``: always re-derive — used under hipGraph capture, where the cast must be part of the replayed work)."""
    # a parameter re-homed into a flat buffer (engine/flat.py) is a `.data` view of it: the fused optimizer / FlatEMA mutate the
    # FLAT tensor, which bumps the flat buffer's version counter, never the parameter's own — key on both
    owner = getattr(w, "_sy11_owner", None)
    key = (w._version, -1 if owner is None else owner._version, w.data_ptr(), dtype, w.device)
    cache = holder.__dict__.get("_sy11_wcache")
    if not force and cache is not None and cache[0] == key:
        return cache[1]
    k = ops.filter_krsc(w.detach())
    if k.dtype != dtype:
        k = k.to(dtype)
    if not force:
        holder.__dict__["_sy11_wcache"] = (key, k)
    return k


def conv2d_bias_run(ec: Ctx, conv: nn.Conv2d, x: Act, out: Act) -> Act:
    """Bare nn.Conv2d with bias (Detect's last 1x1 convs, head.py:44-55): writes f32 logits into ``out``."""
    k, s, p, d = _int(conv.kernel_size), _int(conv.stride), _int(conv.padding), _int(conv.dilation)
    w = ec.w_views.get(id(conv.weight)) if ec.w_views else None
    if w is None:
        w = working_filter(conv, conv.weight, ec.dtype, ec.capturing)
    bias = conv.bias.detach().float() if conv.bias is not None else None
    out_f32 = out.data.dtype == torch.float32 and ec.dtype != torch.float32
    ops.conv2d_fwd(x.data, w, out.data, k, s, p, d, 1, bias=bias, out_f32=out_f32)
    if ec.record:
        def bw():
            dz = out.grad_read()
            gs = ec.grads
            n = dz.shape[3]
            has_bias = conv.bias is not None and id(conv.bias) in gs.views
            epc = 16 // torch.empty((), dtype=ec.dtype).element_size()
            npad = -(-n // epc) * epc
            if npad == n and dz.dtype == ec.dtype and (dz.stride(2) * dz.element_size()) % 16 == 0:
                dy, wk = dz, w                         # the gradient already is a compute-dtype operand (f32 models)
                if has_bias:
                    one = torch.ones(n, device=ec.device)
                    zero = torch.zeros(n, device=ec.device)
                    scratch = torch.empty(n, device=ec.device)
                    ops.bn_act_bwd_reduce(dz, dz, zero, one, one, zero, False, gs.grad_vec(conv.bias), scratch)
            elif dz.dtype == torch.float32 and npad <= 256:
                # f32 logit gradient -> 16-bit operand (channel count padded to a 16-byte multiple, e.g. nc = 2 -> 8) and the bias
                # gradient, in ONE pass (sy11_bias_grad_cast)
                dy = torch.empty((*dz.shape[:3], npad), dtype=ec.dtype, device=ec.device)
                ops.bias_grad_cast(dz, dy, gs.grad_vec(conv.bias) if has_bias else None,
                                   torch.empty((512, n), dtype=torch.float32, device=ec.device) if has_bias else None)
                wk = w if npad == n else torch.zeros((npad, *w.shape[1:]), dtype=w.dtype, device=w.device)
                if npad != n:
                    wk[:n].copy_(w)
            else:
                if has_bias:
                    one = torch.ones(n, device=ec.device)
                    zero = torch.zeros(n, device=ec.device)
                    scratch = torch.empty(n, device=ec.device)
                    ops.bn_act_bwd_reduce(dz, dz, zero, one, one, zero, False, gs.grad_vec(conv.bias), scratch)
                dy = torch.zeros((*dz.shape[:3], npad), dtype=ec.dtype, device=ec.device)
                dy[..., :n].copy_(dz)
                wk = w if npad == n else torch.zeros((npad, *w.shape[1:]), dtype=w.dtype, device=w.device)
                if npad != n:
                    wk[:n].copy_(w)
            if id(conv.weight) in gs.views:
                if npad == n:
                    ops.conv2d_wgrad(x.data, dy, gs.grad_krsc(conv.weight), k, s, p, d, 1)
                else:
                    dwp = torch.zeros((npad, *w.shape[1:]), dtype=torch.float32, device=ec.device)
                    ops.conv2d_wgrad(x.data, dy, dwp, k, s, p, d, 1)
                    gs.grad_krsc(conv.weight).add_(dwp[:n])
            if x.req:
                g, acc = x.grad_for_write()
                wtr = ec.transposed(conv.weight, wk) if wk is w else ops.weight_transpose(wk)
                ops.conv2d_dgrad(dy, wtr, g, tuple(dy.shape), k, s, p, d, 1, accumulate=acc)
        ec.tape.append(bw)
    return out


class Conv(nn.Module):
    """Conv2d + BatchNorm2d + SiLU: args (ch_in, ch_out, kernel, stride, padding, groups, dilation, activation)."""

    default_act = nn.SiLU()  # shared instance, rebindable by a YAML `activation:` key (tasks.py:979-980)

    def __init__(self, c1, c2, k=1, s=1, p=None, g=1, d=1, act=True):
        super().__init__()
        self.conv = nn.Conv2d(c1, c2, k, s, autopad(k, p, d), groups=g, dilation=d, bias=False)
        self.bn = nn.BatchNorm2d(c2)
        self.act = self.default_act if act is True else act if isinstance(act, nn.Module) else nn.Identity()
        self.conv.weight.data = self.conv.weight.data.contiguous(memory_format=torch.channels_last)
        self.conv.weight._sy11_groups = g

    # ---- torch-facing API (tensor in / tensor out, differentiable)
    def forward(self, x):
        return run_module(self, x)[0]

    def forward_fuse(self, x):
        return run_module(self, x)[0]

    # ---- engine
    def _silu(self) -> bool:
        if isinstance(self.act, nn.SiLU):
            return True
        if isinstance(self.act, nn.Identity):
            return False
        raise ops._lib.Sy11Error(f"activation {type(self.act).__name__} has no HIP kernel (SiLU / Identity only)")

    def _run(self, ec: Ctx, x: Act, out: Act = None, res: Act = None) -> Act:
        conv = self.conv
        k, s, p, d, g = _int(conv.kernel_size), _int(conv.stride), _int(conv.padding), _int(conv.dilation), conv.groups
        B, H, W, C1 = x.shape
        N = conv.out_channels
        OH, OW = ops.conv_out_hw(H, W, k, s, p, d)
        silu = self._silu()
        w = ec.w_views.get(id(conv.weight)) if ec.w_views else None
        if w is None:
            w = working_filter(conv, conv.weight, ec.dtype, ec.capturing)
        stem = x.raw is not None and C1 == 3 and k == 3 and g == 1 and d == 1 and N <= 64
        # input channel count that is not a whole number of 16-byte vectors (the 3-channel image in front of a stem wider than
        # 64, e.g. yolo11x: 96): run the dense kernels on zero-padded channels — zeros contribute nothing to y, dw gets sliced
        epc = 16 // torch.empty((), dtype=ec.dtype).element_size()
        padded = (not stem) and g == 1 and C1 % epc != 0
        x_real, w_real = x, w
        if padded:
            Cp = -(-C1 // epc) * epc
            xp = torch.zeros((B, H, W, Cp), dtype=ec.dtype, device=ec.device)
            xp[..., :C1].copy_(x.data)
            wp = torch.zeros((N, k, k, Cp), dtype=w.dtype, device=w.device)
            wp[..., :C1].copy_(w)
            x, w = Act(xp, req=False), wp
        if out is None:
            out = Act(ec.empty(B, OH, OW, N))
        bn = getattr(self, "bn", None)
        if bn is None:                                   # fused Conv (BaseModel.fuse): conv + bias + act in one kernel
            if ec.record and conv.weight.requires_grad:
                raise ops._lib.Sy11Error("a fused Conv is inference-only (fuse_conv_and_bn makes it requires_grad=False)")
            bias = conv.bias.detach().float() if conv.bias is not None else None
            if stem:
                ops.stem_conv_fwd(x.raw, w, out.data, s, p, bias=bias, silu=silu)
            else:
                ops.conv2d_fwd(x.data, w, out.data, k, s, p, d, g, bias=bias, silu=silu)
            if res is not None:
                ops.copy2d(res.data, out.data, accumulate=True)
            return out
        y = ec.empty(B, OH, OW, N)
        if not ec.training:                              # eval with running statistics
            if ec.record:
                raise ops._lib.Sy11Error("gradients through eval-mode BatchNorm are not implemented; call model.train()")
            scale = (bn.weight.detach() / torch.sqrt(bn.running_var + bn.eps)).float()
            shift = (bn.bias.detach() - bn.running_mean * scale).float()
            if stem:
                ops.stem_conv_fwd(x.raw, w, y, s, p)
            else:
                ops.conv2d_fwd(x.data, w, y, k, s, p, d, g)
            ops.bn_act_fwd(y, scale, shift, out.data, silu=silu, res=res.data if res is not None else None)
            return out
        slots = _BN_STAT_SLOTS if B * OH * OW >= 128 * 64 else 1        # spread the per-channel stat atomics (see sy11.h)
        st = ec.zeros(2, slots, N)
        v = torch.empty((4, N), dtype=torch.float32, device=ec.device)
        mean, rstd, scale, shift = v[0], v[1], v[2], v[3]
        gamma, beta = bn.weight.detach().float(), bn.bias.detach().float()
        mom = 0.1 if bn.momentum is None else bn.momentum
        track = bn.track_running_stats and bn.running_mean is not None
        rm, rv = (bn.running_mean, bn.running_var) if track else (None, None)
        if _BN_TAIL:
            # conv + batch statistics + their finalisation in ONE launch (the last workgroup folds the statistic slots,
            # csrc/bn_tail.h).  Measured r01: 27.7 ms/step vs 26.4 with the separate 5 us finalize kernel — a single workgroup
            # folding 32 slots x C through device-scope loads is slower than C/32 workgroups doing it in parallel.  Off by default.
            tail = (B * OH * OW, gamma, beta, bn.eps, mom, rm, rv, mean, rstd, scale, shift, ec.zeros(max(g, 1)))
            if stem:
                ops.stem_conv_fwd_bn(x.raw, w, y, s, p, (st[0], st[1]), tail)
            else:
                ops.conv2d_fwd_bn(x.data, w, y, k, s, p, d, g, (st[0], st[1]), tail)
        else:
            if stem:
                ops.stem_conv_fwd(x.raw, w, y, s, p, stats=(st[0], st[1]))
            else:
                ops.conv2d_fwd(x.data, w, y, k, s, p, d, g, stats=(st[0], st[1]))
            ops.bn_finalize(B * OH * OW, st[0], st[1], gamma, beta, bn.eps, mom, rm, rv, mean, rstd, scale, shift)
        if track and bn.num_batches_tracked is not None:
            ec.bn_counters.append(bn.num_batches_tracked)
        ops.bn_act_fwd(y, scale, shift, out.data, silu=silu, res=res.data if res is not None else None)
        if ec.record:
            def bw():
                gs = ec.grads
                dz = out.grad_read()
                sg = ec.zeros(2, 8 if B * OH * OW >= 8192 else 1, N)     # 8 slots: spread the same-line atomics
                ops.bn_act_bwd_reduce(y, dz, mean, rstd, scale, shift, silu, sg[0], sg[1])
                dy = torch.empty_like(y)
                has_bn = id(bn.weight) in gs.views
                rg, acc = res.grad_for_write() if (res is not None and res.req) else (None, False)
                # the residual operand's gradient is dz itself: written / added by the apply pass from the values it holds
                ops.bn_act_bwd_apply(y, dz, mean, rstd, scale, shift, gamma, silu, sg[0], sg[1], dy,
                                     gs.grad_vec(bn.weight) if has_bn else None, gs.grad_vec(bn.bias) if has_bn else None,
                                     res_grad=rg, res_accumulate=acc)
                if id(conv.weight) in gs.views:
                    dw = gs.grad_krsc(conv.weight)
                    if stem:
                        ec.on_side(lambda: ops.stem_conv_wgrad(x.raw, dy, dw, s, p), x.raw, dy)
                    elif padded:
                        dwp = torch.zeros(tuple(w.shape), dtype=torch.float32, device=ec.device)
                        ops.conv2d_wgrad(x.data, dy, dwp, k, s, p, d, g)
                        dw.add_(dwp[..., :C1])
                    else:
                        xd = x.data
                        ec.on_side(lambda: ops.conv2d_wgrad(xd, dy, dw, k, s, p, d, g), xd, dy)
                if padded and x_real.req:
                    gp = torch.zeros_like(x.data)
                    ops.conv2d_dgrad(dy, ops.weight_transpose(w), gp, (B, OH, OW, N), k, s, p, d, 1, accumulate=True)
                    gx, acc = x_real.grad_for_write()
                    if acc:
                        gx.add_(gp[..., :C1])
                    else:
                        gx.copy_(gp[..., :C1])
                if x.req:
                    gx, acc = x.grad_for_write()
                    if g == 1 or g != C1 or g != N:          # dense, or grouped as `g` dense slices (DDWConv g = 8)
                        if ops.dgrad_leaves_holes(k, s, p, d) and not acc:
                            gx.zero_()
                            acc = True
                        wtr = ec.transposed(conv.weight, w) if g == 1 else ops.weight_transpose(w, g)
                        ops.conv2d_dgrad(dy, wtr, gx, (B, OH, OW, N), k, s, p, d, g, accumulate=acc)
                    else:                                    # depthwise: the kernel reads the filter as stored
                        ops.conv2d_dgrad(dy, w, gx, (B, OH, OW, N), k, s, p, d, g, accumulate=acc)
            ec.tape.append(bw)
        return out


class DWConv(Conv):
    """Depth-wise convolution (conv.py:687-692)."""

    def __init__(self, c1, c2, k=1, s=1, d=1, act=True):
        super().__init__(c1, c2, k, s, g=math.gcd(c1, c2), d=d, act=act)


class Concat(nn.Module):
    """Concatenate a list of tensors along ``dimension`` (conv.py:1810-1821); channel concat runs as strided copies."""

    def __init__(self, dimension=1):
        super().__init__()
        self.d = dimension

    def forward(self, x):
        return run_module(self, x)[0]

    def _run(self, ec: Ctx, xs) -> Act:
        if self.d != 1:
            raise ops._lib.Sy11Error("only channel concatenation (dimension=1) is on the hot path")
        B, H, W, _ = xs[0].shape
        out = Act(ec.empty(B, H, W, sum(a.C for a in xs)))
        c = 0
        parts = []
        for a in xs:
            sl = out.slice(c, c + a.C)
            ops.copy2d(a.data, sl.data)
            parts.append((a, sl))
            c += a.C
        if ec.record:
            def bw():
                for a, sl in parts:
                    if a.req:
                        g, acc = a.grad_for_write()
                        ops.copy2d(sl.grad_read(), g, accumulate=acc)
            ec.tape.append(bw)
        return out


class DDWConv(nn.Module):
    """Fusion-variant down-sampler (conv.py:694-710): grouped (g = 8, optionally dilated) k x k Conv, then a 1x1 Conv.
    Args (ch_in, ch_out, kernel, stride, dilation, activation)."""

    def __init__(self, c1, c2, k=3, s=2, d=1, act=True):
        super().__init__()
        self.conv1 = Conv(c1, c2, k, s, g=8, d=d, act=act)
        self.kz = k
        self.conv2 = Conv(c2, c2, k=1, s=1)

    def forward(self, x):
        return run_module(self, x)[0]

    def _run(self, ec: Ctx, x: Act, out: Act = None) -> Act:
        return self.conv2._run(ec, self.conv1._run(ec, x), out=out)


class WeightedSpatialAttention(nn.Module):
    """x * sigmoid(conv_k([mean_c x, max_c x])) (conv.py:1839-1852).  Parameter holder for ``cv1`` (2 -> 1, no bias);
    inside Fusion the 3x3 map is computed by sy11_sab_map_fwd."""

    def __init__(self, kernel_size=7):
        super().__init__()
        assert kernel_size in {3, 7}, "kernel size must be 3 or 7"
        self.cv1 = nn.Conv2d(2, 1, kernel_size, padding=kernel_size // 2, bias=False)
        self.act = nn.Sigmoid()


class GCT(nn.Module):
    """Gated channel transformation (conv.py:2284-2301): x * (1 + tanh(e * gamma / rms_c(e) + beta)), e = ||x||_hw * alpha.
    Parameter holder; the gate vector is computed by sy11_gct_gate_fwd."""

    def __init__(self, num_channels, epsilon=1e-5, mode="l2", after_relu=False):
        super().__init__()
        self.alpha = nn.Parameter(torch.ones(1, num_channels, 1, 1))
        self.gamma = nn.Parameter(torch.zeros(1, num_channels, 1, 1))
        self.beta = nn.Parameter(torch.zeros(1, num_channels, 1, 1))
        self.epsilon = epsilon
        self.mode = mode
        self.after_relu = after_relu


def _vec(p: torch.Tensor) -> torch.Tensor:
    """Flat f32 view of a (1, C, 1, 1) parameter (or of its gradient view)."""
    v = p.detach().reshape(-1)
    return v if v.dtype == torch.float32 else v.float()


class Fusion(nn.Module):
    """Fusion(inc_list, 'ESChannel', c1=128) (conv.py:1854-1857, 1928-1931, 2087-2127; parse_model forces the
    'ESChannel' branch, tasks.py:1132-1135):  sum_i [ chunk_i(GCT(cat(x))) + SAB(x_i) ]
    = sum_i x_i * (gate[b, i*C + c] + S_i[b, h, w]) for 2 or 3 inputs of c1 channels."""

    def __init__(self, inc_list, fusion="bifpn", c1=128):
        super().__init__()
        if fusion != "ESChannel":
            raise ops._lib.Sy11Error(f"Fusion('{fusion}') has no HIP kernel: only the 'ESChannel' branch the model parser selects")
        self.fusion = fusion
        self.sab = WeightedSpatialAttention(3)
        self.gsc2 = GCT(c1 * 2)
        self.gsc3 = GCT(c1 * 3)

    def forward(self, x):
        return run_module(self, x)[0]

    def _run(self, ec: Ctx, xs, out: Act = None) -> Act:
        n = len(xs)
        if n not in (2, 3):
            raise ops._lib.Sy11Error("Fusion('ESChannel') takes 2 or 3 inputs")
        B, H, W, Cn = xs[0].shape
        gct = self.gsc2 if n == 2 else self.gsc3
        if gct.alpha.numel() != n * Cn or any(tuple(a.shape) != (B, H, W, Cn) for a in xs):
            raise ops._lib.Sy11Error(f"Fusion: inputs must all be (B,{gct.alpha.numel() // n},H,W)")
        dev = ec.device
        f32 = dict(dtype=torch.float32, device=dev)
        sq = ec.zeros(B, n * Cn)
        mm = [torch.empty((B, H, W, 2), **f32) for _ in range(n)]
        am = [torch.empty((B, H, W), dtype=torch.int16, device=dev) for _ in range(n)]
        S = [torch.empty((B, H, W), **f32) for _ in range(n)]
        w18 = ops.filter_krsc(self.sab.cv1.weight.detach()).float().reshape(-1)
        for i, a in enumerate(xs):
            ops.fusion_stats(a.data, mm[i], am[i], sq[:, i * Cn:(i + 1) * Cn])
            ops.sab_map_fwd(mm[i], w18, S[i])
        alpha, gamma, beta = _vec(gct.alpha), _vec(gct.gamma), _vec(gct.beta)
        G = torch.empty((B, n * Cn), **f32)
        ops.gct_gate_fwd(sq, alpha, gamma, beta, gct.epsilon, G)
        if out is None:
            out = Act(ec.empty(B, H, W, Cn))
        ops.fusion_combine([a.data for a in xs], S, G, out.data)
        if ec.record:
            def bw():
                gs = ec.grads
                dout = out.grad_read()
                dG = ec.zeros(B, n * Cn)
                dS = [torch.empty((B, H, W), **f32) for _ in range(n)]
                dmm = [torch.empty((B, H, W, 2), **f32) for _ in range(n)]
                has = lambda p: id(p) in gs.views                       # noqa: E731
                scratch = torch.zeros(18 + 3 * n * Cn, **f32)
                dw18 = gs.grad_krsc(self.sab.cv1.weight).reshape(-1) if has(self.sab.cv1.weight) else scratch[:18]
                for i, a in enumerate(xs):
                    ops.fusion_bwd_reduce(dout, a.data, dG[:, i * Cn:(i + 1) * Cn], dS[i])
                    ops.sab_map_bwd(dS[i], S[i], mm[i], w18, dmm[i], dw18)
                q = torch.empty((B, n * Cn), **f32)
                pg = [gs.grad_vec(p).reshape(-1) if has(p) else scratch[18 + j * n * Cn:18 + (j + 1) * n * Cn]
                      for j, p in enumerate((gct.alpha, gct.gamma, gct.beta))]
                ops.gct_gate_bwd(sq, alpha, gamma, beta, gct.epsilon, dG, q, pg[0], pg[1], pg[2])
                for i, a in enumerate(xs):
                    if a.req:
                        g, acc = a.grad_for_write()
                        ops.fusion_bwd_apply(dout, a.data, G[:, i * Cn:(i + 1) * Cn], q[:, i * Cn:(i + 1) * Cn], S[i], dmm[i], am[i], g, acc)
            ec.tape.append(bw)
        return out
