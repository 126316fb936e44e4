"""Block modules with the reference's names / signatures / state_dict keys, executed by libsy11
(ultralytics/nn/modules/block.py: DFL :65-83, SPPF :179-198, C2f :444-471, C3 :490-504, Bottleneck :713-726,
C3k2 :1659-1671, C3k :1672-1680, Attention :1878-1933, PSABlock :1973-2007, C2PSA :2100-2139).

Every ``torch.cat`` / ``chunk`` / ``split`` of the reference is a channel slice of one NHWC buffer here: producers
write into their slice, consumers read theirs, gradients accumulate into the matching slice (concat by pointer).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from ... import ops
from ...engine import Act, Ctx, run_module
from .conv import Conv

__all__ = ("DFL", "SPPF", "C2f", "C3", "C3k", "C3k2", "Bottleneck", "Attention", "PSABlock", "C2PSA")


class _EngineModule(nn.Module):
    def forward(self, x):
        return run_module(self, x)[0]


class DFL(nn.Module):
    """Integral module of Distribution Focal Loss; frozen arange(c1) weight kept for checkpoint parity.  The
    expectation itself is part of the fused Detect decode kernel (sy11_detect_decode)."""

    def __init__(self, c1=16):
        super().__init__()
        self.conv = nn.Conv2d(c1, 1, 1, bias=False).requires_grad_(False)
        self.conv.weight.data[:] = torch.arange(c1, dtype=torch.float).view(1, c1, 1, 1)
        self.c1 = c1


class Bottleneck(_EngineModule):
    """Standard bottleneck: x + cv2(cv1(x)) when shortcut and c1 == c2."""

    def __init__(self, c1, c2, shortcut=True, g=1, k=(3, 3), e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, k[0], 1)
        self.cv2 = Conv(c_, c2, k[1], 1, g=g)
        self.add = shortcut and c1 == c2

    def _run(self, ec: Ctx, x: Act, out: Act = None) -> Act:
        h = self.cv1._run(ec, x)
        return self.cv2._run(ec, h, out=out, res=x if self.add else None)     # residual fused into the BN/SiLU pass


class C2f(_EngineModule):
    """CSP bottleneck with 2 convolutions; all branches live in ONE (2+n)*c-channel buffer."""

    def __init__(self, c1, c2, n=1, shortcut=False, g=1, e=0.5):
        super().__init__()
        self.c = int(c2 * e)
        self.cv1 = Conv(c1, 2 * self.c, 1, 1)
        self.cv2 = Conv((2 + n) * self.c, c2, 1)
        self.m = nn.ModuleList(Bottleneck(self.c, self.c, shortcut, g, k=((3, 3), (3, 3)), e=1.0) for _ in range(n))
        self.n = n

    def _run(self, ec: Ctx, x: Act, out: Act = None) -> Act:
        B, H, W, _ = x.shape
        c, n = self.c, len(self.m)
        cat = Act(ec.empty(B, H, W, (2 + n) * c))
        self.cv1._run(ec, x, out=cat.slice(0, 2 * c))
        for j, m in enumerate(self.m):
            m._run(ec, cat.slice((1 + j) * c, (2 + j) * c), out=cat.slice((2 + j) * c, (3 + j) * c))
        return self.cv2._run(ec, cat, out=out)

    forward_split = _EngineModule.forward


class C3(_EngineModule):
    """CSP bottleneck with 3 convolutions: cv3(cat(m(cv1(x)), cv2(x)))."""

    def __init__(self, c1, c2, n=1, shortcut=True, g=1, e=0.5):
        super().__init__()
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c1, c_, 1, 1)
        self.cv3 = Conv(2 * c_, c2, 1)
        self.m = nn.Sequential(*(Bottleneck(c_, c_, shortcut, g, k=((1, 1), (3, 3)), e=1.0) for _ in range(n)))

    def _run(self, ec: Ctx, x: Act, out: Act = None) -> Act:
        B, H, W, _ = x.shape
        c_ = self.cv1.conv.out_channels
        cat = Act(ec.empty(B, H, W, 2 * c_))
        h = self.cv1._run(ec, x) if len(self.m) else self.cv1._run(ec, x, out=cat.slice(0, c_))
        for j, m in enumerate(self.m):
            h = m._run(ec, h, out=cat.slice(0, c_) if j == len(self.m) - 1 else None)
        self.cv2._run(ec, x, out=cat.slice(c_, 2 * c_))
        return self.cv3._run(ec, cat, out=out)


class C3k(C3):
    """C3 with k x k bottlenecks at e=1.0."""

    def __init__(self, c1, c2, n=1, shortcut=True, g=1, e=0.5, k=3):
        super().__init__(c1, c2, n, shortcut, g, e)
        c_ = int(c2 * e)
        self.m = nn.Sequential(*(Bottleneck(c_, c_, shortcut, g, k=(k, k), e=1.0) for _ in range(n)))


class C3k2(C2f):
    """C2f whose inner blocks are C3k (c3k=True) or plain Bottlenecks."""

    def __init__(self, c1, c2, n=1, c3k=False, e=0.5, g=1, shortcut=True):
        super().__init__(c1, c2, n, shortcut, g, e)
        self.m = nn.ModuleList(
            C3k(self.c, self.c, 2, shortcut, g) if c3k else Bottleneck(self.c, self.c, shortcut, g) for _ in range(n)
        )


class SPPF(_EngineModule):
    """Spatial pyramid pooling (fast): cv1 -> 3 cascaded 5x5 max-pools -> cv2 over the 4-way concat buffer."""

    def __init__(self, c1, c2, k=5):
        super().__init__()
        c_ = c1 // 2
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c_ * 4, c2, 1, 1)
        self.m = nn.MaxPool2d(kernel_size=k, stride=1, padding=k // 2)

    def _run(self, ec: Ctx, x: Act, out: Act = None) -> Act:
        if self.m.kernel_size != 5:
            raise ops._lib.Sy11Error("SPPF: only the 5x5 pool has a HIP kernel")
        B, H, W, _ = x.shape
        c_ = self.cv1.conv.out_channels
        cat = Act(ec.empty(B, H, W, 4 * c_))
        parts = [cat.slice(i * c_, (i + 1) * c_) for i in range(4)]
        self.cv1._run(ec, x, out=parts[0])
        idx = [torch.empty((B, H, W, c_), dtype=torch.uint8, device=ec.device) if ec.record else None for _ in range(3)]
        for i in range(3):
            ops.maxpool5_fwd(parts[i].data, parts[i + 1].data, idx[i])
        if ec.record:
            def bw():       # appended BEFORE cv2 runs, so in the reversed tape it fires right after cv2's backward
                for i in (2, 1, 0):
                    g, acc = parts[i].grad_for_write()
                    ops.maxpool5_bwd(parts[i + 1].grad_read(), idx[i], g, accumulate=acc)
            ec.tape.append(bw)
        return self.cv2._run(ec, cat, out=out)


class Attention(_EngineModule):
    """Multi-head self-attention over the H*W tokens with a depth-wise positional branch."""

    def __init__(self, dim, num_heads=8, attn_ratio=0.5):
        super().__init__()
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.key_dim = int(self.head_dim * attn_ratio)
        self.scale = self.key_dim**-0.5
        nh_kd = self.key_dim * num_heads
        h = dim + nh_kd * 2
        self.qkv = Conv(dim, h, 1, act=False)
        self.proj = Conv(dim, dim, 1, act=False)
        self.pe = Conv(dim, dim, 3, 1, g=dim, act=False)

    def _run(self, ec: Ctx, x: Act, out: Act = None, res: Act = None) -> Act:
        B, H, W, Cn = x.shape
        N = H * W
        nh, kd, hd = self.num_heads, self.key_dim, self.head_dim
        qkv = self.qkv._run(ec, x)                       # (B,H,W, nh*(2kd+hd)), per head [q k v]
        # v of every head gathered into one C-channel tensor for the depth-wise `pe` branch (v.reshape(B,C,H,W))
        hc = 2 * kd + hd
        v = Act(ec.empty(B, H, W, Cn))
        for h in range(nh):
            ops.copy2d(qkv.slice(h * hc + 2 * kd, (h + 1) * hc).data, v.slice(h * hd, (h + 1) * hd).data)
        o = Act(ec.empty(B, H, W, Cn))
        p = torch.empty((B, nh, N, N), dtype=torch.float32, device=ec.device)
        ops.attention_fwd(qkv.data, nh, kd, hd, o.data, p)
        if ec.record:
            def bw_attn():
                # o.grad holds d(attn_out + pe(v)); attention core backward writes dq, dk, dv for all heads
                dqkv, acc = qkv.grad_for_write()
                assert not acc
                ops.attention_bwd(qkv.data, nh, kd, hd, p, o.grad_read(), dqkv, o=o.data)
                gv = v.grad_read()                       # from the pe branch
                for h in range(nh):
                    ops.copy2d(gv[..., h * hd:(h + 1) * hd], dqkv[..., h * hc + 2 * kd:(h + 1) * hc], accumulate=True)
            ec.tape.append(bw_attn)
        # x = attn + pe(v): pe's BN pass adds `o` as its residual, result feeds proj
        s = self.pe._run(ec, v, res=o)
        return self.proj._run(ec, s, out=out, res=res)


class PSABlock(_EngineModule):
    """x + attn(x), then x + ffn(x)."""

    def __init__(self, c, attn_ratio=0.5, num_heads=4, shortcut=True) -> None:
        super().__init__()
        self.attn = Attention(c, attn_ratio=attn_ratio, num_heads=num_heads)
        self.ffn = nn.Sequential(Conv(c, c * 2, 1), Conv(c * 2, c, 1, act=False))
        self.add = shortcut

    def _run(self, ec: Ctx, x: Act, out: Act = None) -> Act:
        a = self.attn._run(ec, x, res=x if self.add else None)
        f = self.ffn[0]._run(ec, a)
        return self.ffn[1]._run(ec, f, out=out, res=a if self.add else None)


class C2PSA(_EngineModule):
    """cv1 -> split (a, b) -> n x PSABlock(b) -> cv2(cat(a, b))."""

    def __init__(self, c1, c2, n=1, e=0.5):
        super().__init__()
        assert c1 == c2
        self.c = int(c1 * e)
        self.cv1 = Conv(c1, 2 * self.c, 1, 1)
        self.cv2 = Conv(2 * self.c, c1, 1)
        self.m = nn.Sequential(*(PSABlock(self.c, attn_ratio=0.5, num_heads=self.c // 64) for _ in range(n)))

    def _run(self, ec: Ctx, x: Act, out: Act = None) -> Act:
        B, H, W, _ = x.shape
        c = self.c
        ab = self.cv1._run(ec, x)
        cat = Act(ec.empty(B, H, W, 2 * c))
        ops.copy2d(ab.slice(0, c).data, cat.slice(0, c).data)
        b = ab.slice(c, 2 * c)
        for j, m in enumerate(self.m):
            b = m._run(ec, b, out=cat.slice(c, 2 * c) if j == len(self.m) - 1 else None)
        if not len(self.m):
            ops.copy2d(b.data, cat.slice(c, 2 * c).data)
        if ec.record:
            a_src, a_dst = ab.slice(0, c), cat.slice(0, c)

            def bw():
                g, acc = a_src.grad_for_write()
                ops.copy2d(a_dst.grad_read(), g, accumulate=acc)
            # must run after cv2's backward (which fills cat.grad): place it BEFORE cv2 on the tape
            ec.tape.append(bw)
        return self.cv2._run(ec, cat, out=out)
