"""Tensor-level wrappers over the C-ABI (sy11._lib): build descriptors from torch tensors, pass raw device
pointers + the current HIP stream.  torch is plumbing only (memory, streams); every computation is a libsy11 kernel.

An NHWC *view* is a torch tensor of shape (B, H, W, C) whose strides are (H*W*ld, W*ld, ld, 1): a channel slice
``buf[..., a:b]`` of a contiguous NHWC buffer qualifies (ld = buf.shape[-1]) — concat by pointer.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import EPI_ACCUM, EPI_OUT_F32, EPI_SILU, BnTail, ConvDesc, OptDesc, call

_DT = {torch.float32: _lib.F32, torch.float16: _lib.F16, torch.bfloat16: _lib.BF16}


def dt_code(t: torch.dtype) -> int:
    try:
        return _DT[t]
    except KeyError:
        raise _lib.Sy11Error(f"unsupported dtype {t}") from None


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _need_gpu(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.Sy11Error("sy11 ops need tensors on the MI355X (cuda) device; there is no CPU path")


def view_ld(t: torch.Tensor) -> int:
    """Pixel stride of an NHWC view; validates the stride pattern."""
    if t.dim() != 4:
        raise _lib.Sy11Error(f"expected a 4-d NHWC view, got shape {tuple(t.shape)}")
    B, H, W, Cn = t.shape
    ld = t.stride(2) if W > 1 else (t.stride(1) // max(W, 1) if H > 1 else max(t.stride(0) // max(H * W, 1), Cn))
    ok = (Cn == 1 or t.stride(3) == 1) and (W == 1 or t.stride(2) == ld) and (H == 1 or t.stride(1) == W * ld) \
        and (B == 1 or t.stride(0) == H * W * ld) and ld >= Cn
    if not ok:
        raise _lib.Sy11Error(f"not an NHWC view: shape {tuple(t.shape)} strides {t.stride()}")
    return ld


def nhwc_empty(B, H, W, Cn, dtype, device):
    return torch.empty((B, H, W, Cn), dtype=dtype, device=device)


def dgrad_leaves_holes(k, s, p, d=1) -> bool:
    """True when some input pixels receive no tap at all (stride > reach, or stride and dilation sharing a factor):
    sy11_conv2d_dgrad then needs a zeroed dx and SY11_EPI_ACCUM."""
    return any(all((ph + p - r * d) % s for r in range(k)) for ph in range(s))


def conv_out_hw(H, W, k, s, p, d=1):
    return (H + 2 * p - d * (k - 1) - 1) // s + 1, (W + 2 * p - d * (k - 1) - 1) // s + 1


def _desc(x_shape, x_ld, y_shape, y_ld, dtype, k, s, p, d, groups, flags, slots=1):
    B, IH, IW, Cn = x_shape
    _, OH, OW, N = y_shape
    kh, kw = (k, k) if isinstance(k, int) else k
    if _lib.PROFILE is not None:     # algorithmic FLOPs of this conv problem (same for fwd / dgrad / wgrad)
        _lib.PROFILE_META = {"flops": 2.0 * B * OH * OW * N * (Cn // groups) * kh * kw, "groups": groups,
                             "bytes": (B * IH * IW * Cn + B * OH * OW * N + N * (Cn // groups) * kh * kw)
                             * torch.empty((), dtype=dtype).element_size(),
                             "desc": f"B{B} {IH}x{IW}x{Cn}->{OH}x{OW}x{N} k{kh} s{s} g{groups}"}
    return ConvDesc(dt_code(dtype), B, IH, IW, Cn, x_ld, OH, OW, N, y_ld, kh, kw, s, s, p, p, d, d, groups, flags, slots)


def filter_krsc(w: torch.Tensor) -> torch.Tensor:
    """OIHW parameter -> memory [O][KH][KW][I].  Free when the parameter is kept in channels_last format."""
    v = w.permute(0, 2, 3, 1)
    return v if v.is_contiguous() else v.contiguous()


def conv2d_fwd(x, w_krsc, y, k, s=1, p=0, d=1, groups=1, bias=None, stats=None, silu=False, out_f32=False):
    """x, y: NHWC views; w_krsc: contiguous [N][KH][KW][C/groups]; stats: (sum, sumsq), each f32 [N] or [slots][N],
    accumulated into."""
    _need_gpu(x, w_krsc, y, bias)
    flags = (EPI_SILU if silu else 0) | (EPI_OUT_F32 if out_f32 else 0)
    slots = stats[0].shape[0] if (stats and stats[0].dim() == 2) else 1
    dsc = _desc(x.shape, view_ld(x), y.shape, view_ld(y), x.dtype, k, s, p, d, groups, flags, slots)
    call("sy11_conv2d_fwd", C.byref(dsc), _p(x), _p(w_krsc), _p(bias), _p(y), _p(stats[0]) if stats else None,
         _p(stats[1]) if stats else None, _stream())
    return y


def _bn_tail(count, gamma, beta, eps, momentum, running_mean, running_var, mean, rstd, scale, shift, ticket):
    return BnTail(_p(gamma), _p(beta), _p(running_mean), _p(running_var), _p(mean), _p(rstd), _p(scale), _p(shift), _p(ticket),
                  float(eps), float(momentum), float(count))


def conv2d_fwd_bn(x, w_krsc, y, k, s, p, d, groups, stats, bn):
    """Train-mode Conv.forward up to the BN statistics: conv + per-channel sum/sumsq + (in the kernel tail) mean / rstd /
    scale / shift / running-stat update.  ``bn`` = (count, gamma, beta, eps, momentum, running_mean, running_var, mean, rstd,
    scale, shift, ticket) with ``ticket`` a zeroed 4-byte-per-group buffer."""
    _need_gpu(x, w_krsc, y)
    slots = stats[0].shape[0] if stats[0].dim() == 2 else 1
    dsc = _desc(x.shape, view_ld(x), y.shape, view_ld(y), x.dtype, k, s, p, d, groups, 0, slots)
    tail = _bn_tail(*bn)
    call("sy11_conv2d_fwd_bn", C.byref(dsc), _p(x), _p(w_krsc), _p(y), _p(stats[0]), _p(stats[1]), C.byref(tail), _stream())
    return y


def stem_conv_fwd_bn(x_nchw, w_krsc, y, s, p, stats, bn):
    _need_gpu(x_nchw, w_krsc, y)
    if x_nchw.dtype != torch.float32 or not x_nchw.is_contiguous() or x_nchw.shape[1] != 3:
        raise _lib.Sy11Error("stem_conv_fwd_bn: x must be a contiguous NCHW f32 image with 3 channels")
    B, _, IH, IW = x_nchw.shape
    slots = stats[0].shape[0] if stats[0].dim() == 2 else 1
    dsc = _desc((B, IH, IW, 3), 3, y.shape, view_ld(y), y.dtype, 3, s, p, 1, 1, 0, slots)
    tail = _bn_tail(*bn)
    call("sy11_stem_conv_fwd_bn", C.byref(dsc), _p(x_nchw), _p(w_krsc), _p(y), _p(stats[0]), _p(stats[1]), C.byref(tail), _stream())
    return y


def weight_transpose(w_krsc: torch.Tensor, groups: int = 1) -> torch.Tensor:
    """dgrad filter: [C][KH][KW][N] for a dense conv; [groups][C/g][KH][KW][N/g] for a grouped one (each group's block
    transposed on its own — the layout sy11_conv2d_dgrad documents)."""
    N, KH, KW, Cn = w_krsc.shape
    if groups > 1:
        ng = N // groups
        wt = torch.empty((groups, Cn, KH, KW, ng), dtype=w_krsc.dtype, device=w_krsc.device)
        for g in range(groups):
            call("sy11_weight_transpose", dt_code(w_krsc.dtype), ng, KH * KW, Cn, _p(w_krsc[g * ng:(g + 1) * ng]), _p(wt[g]), _stream())
        return wt
    wt = torch.empty((Cn, KH, KW, N), dtype=w_krsc.dtype, device=w_krsc.device)
    call("sy11_weight_transpose", dt_code(w_krsc.dtype), N, KH * KW, Cn, _p(w_krsc), _p(wt), _stream())
    return wt


def conv2d_dgrad(dy, w_or_wt, dx, y_shape, k, s=1, p=0, d=1, groups=1, accumulate=False):
    """dx (NHWC view of the conv INPUT shape) = conv^T(dy).  groups==1: pass the tap-transposed filter."""
    _need_gpu(dy, w_or_wt, dx)
    dsc = _desc(dx.shape, view_ld(dx), y_shape, y_shape[-1], dy.dtype, k, s, p, d, groups, EPI_ACCUM if accumulate else 0)
    call("sy11_conv2d_dgrad", C.byref(dsc), _p(dy), view_ld(dy), _p(w_or_wt), _p(dx), _stream())
    return dx


def conv2d_wgrad(x, dy, dw_f32, k, s=1, p=0, d=1, groups=1):
    """dw_f32 ([N][KH][KW][C/groups], f32) += x (*) dy."""
    _need_gpu(x, dy, dw_f32)
    if dw_f32.dtype != torch.float32 or not dw_f32.is_contiguous():
        raise _lib.Sy11Error("conv2d_wgrad: dw must be a contiguous f32 tensor")
    dsc = _desc(x.shape, view_ld(x), dy.shape, view_ld(dy), x.dtype, k, s, p, d, groups, 0)
    call("sy11_conv2d_wgrad", C.byref(dsc), _p(x), _p(dy), view_ld(dy), _p(dw_f32), _stream())
    return dw_f32


def stem_conv_fwd(x_nchw, w_krsc, y, s=2, p=1, bias=None, stats=None, silu=False):
    _need_gpu(x_nchw, w_krsc, y)
    if x_nchw.dtype != torch.float32 or not x_nchw.is_contiguous() or x_nchw.shape[1] != 3:
        raise _lib.Sy11Error("stem_conv_fwd: x must be a contiguous NCHW f32 image with 3 channels")
    B, _, IH, IW = x_nchw.shape
    slots = stats[0].shape[0] if (stats and stats[0].dim() == 2) else 1
    dsc = _desc((B, IH, IW, 3), 3, y.shape, view_ld(y), y.dtype, 3, s, p, 1, 1, EPI_SILU if silu else 0, slots)
    call("sy11_stem_conv_fwd", C.byref(dsc), _p(x_nchw), _p(w_krsc), _p(bias), _p(y), _p(stats[0]) if stats else None,
         _p(stats[1]) if stats else None, _stream())
    return y


def stem_conv_wgrad(x_nchw, dy, dw_f32, s=2, p=1):
    _need_gpu(x_nchw, dy, dw_f32)
    B, _, IH, IW = x_nchw.shape
    dsc = _desc((B, IH, IW, 3), 3, dy.shape, view_ld(dy), dy.dtype, 3, s, p, 1, 1, 0)
    call("sy11_stem_conv_wgrad", C.byref(dsc), _p(x_nchw), _p(dy), view_ld(dy), _p(dw_f32), _stream())
    return dw_f32


def _mc(t):
    B, H, W, Cn = t.shape
    return B * H * W, Cn


def bn_finalize(count, ssum, ssq, gamma, beta, eps, momentum, running_mean, running_var, mean, rstd, scale, shift):
    slots = ssum.shape[0] if ssum.dim() == 2 else 1
    if _lib.PROFILE is not None:
        _lib.PROFILE_META = {"flops": 0.0, "bytes": float(gamma.numel() * 4 * (2 * slots + 8)), "desc": f"bn_finalize {gamma.numel()}"}
    call("sy11_bn_finalize", gamma.numel(), slots, float(count), _p(ssum), _p(ssq), _p(gamma), _p(beta), eps, momentum,
         _p(running_mean), _p(running_var), _p(mean), _p(rstd), _p(scale), _p(shift), _stream())


def _stream_meta(what, M, Cn, *tensors):
    """Algorithmic bytes of a streaming pass (bench.py's roofline leg): every operand read or written once."""
    if _lib.PROFILE is not None:
        _lib.PROFILE_META = {"flops": 0.0, "bytes": float(sum(M * Cn * t.element_size() for t in tensors if t is not None)),
                             "desc": f"{what} {M}x{Cn}"}


def bn_act_fwd(y, scale, shift, z, silu=True, res=None):
    _need_gpu(y, z, res)
    M, Cn = _mc(y)
    _stream_meta("bn_act_fwd", M, Cn, y, z, res)
    call("sy11_bn_act_fwd", dt_code(y.dtype), M, Cn, _p(y), view_ld(y), _p(scale), _p(shift), int(silu), _p(res),
         view_ld(res) if res is not None else 0, _p(z), view_ld(z), _stream())
    return z


def bn_act_bwd_reduce(y, dz, mean, rstd, scale, shift, silu, sum_g, sum_gx):
    M, Cn = _mc(y)
    _stream_meta("bn_bwd_reduce", M, Cn, y, dz)
    slots = sum_g.shape[0] if sum_g.dim() == 2 else 1
    call("sy11_bn_act_bwd_reduce", dt_code(y.dtype), M, Cn, _p(y), view_ld(y), _p(dz), view_ld(dz), _p(mean), _p(rstd),
         _p(scale), _p(shift), int(silu), _p(sum_g), _p(sum_gx), slots, _stream())


def bn_act_bwd_apply(y, dz, mean, rstd, scale, shift, gamma, silu, sum_g, sum_gx, dy, dgamma, dbeta, res_grad=None, res_accumulate=False):
    """``res_grad``: gradient buffer of a residual operand of the layer's output (same pixels, own stride): (= | +=) dz in the same pass."""
    M, Cn = _mc(y)
    _stream_meta("bn_bwd_apply", M, Cn, y, dz, dy)
    slots = sum_g.shape[0] if sum_g.dim() == 2 else 1
    if res_grad is None:
        call("sy11_bn_act_bwd_apply", dt_code(y.dtype), M, Cn, _p(y), view_ld(y), _p(dz), view_ld(dz), _p(mean), _p(rstd),
             _p(scale), _p(shift), _p(gamma), int(silu), _p(sum_g), _p(sum_gx), slots, _p(dy), view_ld(dy), _p(dgamma),
             _p(dbeta), _stream())
        return
    if res_grad.dtype != y.dtype or tuple(res_grad.shape) != tuple(y.shape):
        raise _lib.Sy11Error("bn_act_bwd_apply: the residual gradient must have the layer's shape and dtype")
    call("sy11_bn_act_bwd_apply_res", dt_code(y.dtype), M, Cn, _p(y), view_ld(y), _p(dz), view_ld(dz), _p(mean), _p(rstd),
         _p(scale), _p(shift), _p(gamma), int(silu), _p(sum_g), _p(sum_gx), slots, _p(dy), view_ld(dy), _p(dgamma),
         _p(dbeta), _p(res_grad), view_ld(res_grad), int(bool(res_accumulate)), _stream())


def bias_grad_cast(dz, dy, dbias=None, partials=None):
    """dz: f32 NHWC view (B,H,W,N); dy: contiguous (B,H,W,npad) of the compute dtype, npad >= N (pad channels zeroed);
    dbias: f32 [N] accumulated into.  One launch (two with ``partials``, a [rows][N] scratch selecting the ordered reduction)."""
    _need_gpu(dz, dy, dbias, partials)
    M, N = _mc(dz)
    if dz.dtype != torch.float32 or not dy.is_contiguous() or tuple(dy.shape[:3]) != tuple(dz.shape[:3]) or dy.shape[3] < N:
        raise _lib.Sy11Error("bias_grad_cast: dz must be f32, dy contiguous with the same pixels and >= N channels")
    call("sy11_bias_grad_cast", dt_code(dy.dtype), M, N, dy.shape[3], _p(dz), view_ld(dz), _p(dy), _p(dbias), _p(partials),
         partials.shape[0] if partials is not None else 0, _stream())
    return dy


def copy2d(src, dst, accumulate=False):
    _need_gpu(src, dst)
    M, Cn = _mc(src)
    if tuple(dst.shape) != tuple(src.shape) or dst.dtype != src.dtype:
        raise _lib.Sy11Error(f"copy2d: shape/dtype mismatch {tuple(src.shape)} vs {tuple(dst.shape)}")
    call("sy11_copy2d", dt_code(src.dtype), M, Cn, _p(src), view_ld(src), _p(dst), view_ld(dst), int(accumulate), _stream())
    return dst


def upsample2x_fwd(x, y):
    B, H, W, Cn = x.shape
    call("sy11_upsample2x_fwd", dt_code(x.dtype), B, H, W, Cn, _p(x), view_ld(x), _p(y), view_ld(y), _stream())
    return y


def upsample2x_bwd(dy, dx, accumulate=False):
    B, H, W, Cn = dx.shape
    call("sy11_upsample2x_bwd", dt_code(dx.dtype), B, H, W, Cn, _p(dy), view_ld(dy), _p(dx), view_ld(dx), int(accumulate),
         _stream())
    return dx


def maxpool5_fwd(x, y, idx):
    B, H, W, Cn = x.shape
    call("sy11_maxpool5_fwd", dt_code(x.dtype), B, H, W, Cn, _p(x), view_ld(x), _p(y), view_ld(y), _p(idx), _stream())
    return y


def maxpool5_bwd(dy, idx, dx, accumulate=False):
    B, H, W, Cn = dx.shape
    call("sy11_maxpool5_bwd", dt_code(dx.dtype), B, H, W, Cn, _p(dy), view_ld(dy), _p(idx), _p(dx), view_ld(dx),
         int(accumulate), _stream())
    return dx


def attention_fwd(qkv, heads, kd, hd, o, p):
    B, H, W, _ = qkv.shape
    call("sy11_attention_fwd", dt_code(qkv.dtype), B, H * W, heads, kd, hd, _p(qkv), view_ld(qkv), _p(o), view_ld(o), _p(p),
         _stream())
    return o


def attention_bwd(qkv, heads, kd, hd, p, d_o, dqkv, ws=None, o=None):
    """``o``: the forward output (optional) — with it the 16-bit MFMA path takes softmax's row sums as dO . o and reads P once."""
    B, H, W, _ = qkv.shape
    if ws is None:                                   # caller-owned scratch, sized by the library's own query
        ws = torch.empty(_lib.load().sy11_attention_workspace_bytes(B, H * W, heads) // 4, dtype=torch.float32, device=qkv.device)
    if o is None:
        call("sy11_attention_bwd", dt_code(qkv.dtype), B, H * W, heads, kd, hd, _p(qkv), view_ld(qkv), _p(p), _p(d_o),
             view_ld(d_o), _p(dqkv), view_ld(dqkv), _p(ws), _stream())
    else:
        call("sy11_attention_bwd_o", dt_code(qkv.dtype), B, H * W, heads, kd, hd, _p(qkv), view_ld(qkv), _p(p), _p(o), view_ld(o), _p(d_o),
             view_ld(d_o), _p(dqkv), view_ld(dqkv), _p(ws), _stream())
    return dqkv


def detect_decode(maps, strides, nc):
    """maps: list of contiguous NHWC f32 (B, H, W, 64+nc) -> (B, 4+nc, A) f32."""
    _need_gpu(*maps)
    nl = len(maps)
    B = maps[0].shape[0]
    for m in maps:
        if m.dtype != torch.float32 or not m.is_contiguous() or m.shape[-1] != 64 + nc:
            raise _lib.Sy11Error("detect_decode: maps must be contiguous NHWC f32 with 64+nc channels")
    A = sum(m.shape[1] * m.shape[2] for m in maps)
    out = torch.empty((B, 4 + nc, A), dtype=torch.float32, device=maps[0].device)
    ptrs = (C.c_void_p * nl)(*[m.data_ptr() for m in maps])
    hs = (C.c_int32 * nl)(*[m.shape[1] for m in maps])
    ws = (C.c_int32 * nl)(*[m.shape[2] for m in maps])
    st = (C.c_float * nl)(*[float(s) for s in strides])
    call("sy11_detect_decode", B, nc, nl, C.cast(ptrs, C.c_void_p), C.cast(hs, C.c_void_p), C.cast(ws, C.c_void_p),
         C.cast(st, C.c_void_p), _p(out), _stream())
    return out


def nms_sorted(boxes_sorted: torch.Tensor, iou_thres: float) -> torch.Tensor:
    """boxes sorted by (score desc, index asc), (n,4) f32 contiguous -> bool keep mask (n,)."""
    _need_gpu(boxes_sorted)
    n = boxes_sorted.shape[0]
    keep = torch.zeros((n,), dtype=torch.uint8, device=boxes_sorted.device)
    if n == 0:
        return keep.bool()
    b = boxes_sorted.contiguous().float()
    ws = torch.empty((_lib.load().sy11_nms_workspace_bytes(n) // 8,), dtype=torch.int64, device=b.device)
    call("sy11_nms_sorted", n, _p(b), float(iou_thres), _p(ws), _p(keep), _stream())
    return keep.bool()


def nms_sorted_batched(boxes_sorted: torch.Tensor, counts, iou_thres: float, max_keep: int, chunk_bytes: int = 2 << 30) -> torch.Tensor:
    """Greedy NMS of several images in one go: image i owns counts[i] consecutive rows of ``boxes_sorted`` ((total, 4) f32, each
    image's rows in score-descending order).  -> bool keep mask (total,), at most ``max_keep`` survivors per image.  The bit
    matrices of a launch are bounded by ``chunk_bytes`` (images are processed in as many launches as that takes)."""
    _need_gpu(boxes_sorted)
    total = boxes_sorted.shape[0]
    keep = torch.empty((total,), dtype=torch.uint8, device=boxes_sorted.device)
    if total == 0:
        return keep.bool()
    b = boxes_sorted.contiguous().float()
    lib = _lib.load()
    counts = [int(c) for c in counts]
    need = [c * ((c + 63) // 64) * 8 for c in counts]
    lo, row = 0, 0
    while lo < len(counts):
        hi, size = lo, 0
        while hi < len(counts) and (hi == lo or size + need[hi] <= chunk_bytes):
            size += need[hi]
            hi += 1
        n_rows = sum(counts[lo:hi])
        if n_rows:
            arr = (C.c_int32 * (hi - lo))(*counts[lo:hi])
            ws = torch.empty((max(size // 8, 1),), dtype=torch.int64, device=b.device)
            call("sy11_nms_sorted_batched", hi - lo, C.cast(arr, C.c_void_p), _p(b[row:row + n_rows]), float(iou_thres), int(max_keep), _p(ws),
                 _p(keep[row:row + n_rows]), _stream())
        row += n_rows
        lo = hi
    return keep.bool()


def nms_candidates(pred: torch.Tensor, nc: int, conf_thres: float, multi_label: bool, segment_by_class: bool = False, max_per_image: int = None,
                   class_ok: torch.Tensor = None):
    """Candidates of non_max_suppression from the (B, 4 + nc + nm, A) f32 prediction tensor, in the reference's order (image, anchor,
    class).  -> (key int64 (n,), anchor int32 (n,), class int32 (n,)) on the device, the per-image counts as a host list, and whether
    the keys are segmented by class: key = segment << 32 | ~bits(score), segment = image, or image * nc + class when
    ``segment_by_class`` is asked for AND no image has more than ``max_per_image`` candidates (the caller's `max_nms` cut is per image
    and by score, so it needs the image-wide order) AND the device flag ``class_ok`` (read with the counts) is non-zero.
    One host synchronisation (the counts)."""
    _need_gpu(pred)
    if pred.dtype != torch.float32 or not pred.is_contiguous():
        raise _lib.Sy11Error("nms_candidates: prediction must be a contiguous f32 (B, D, A) tensor")
    B, D, A = pred.shape
    nblk = (A + 255) // 256
    cnt = torch.empty((B, nblk), dtype=torch.int32, device=pred.device)
    call("sy11_nms_candidates", B, D, A, nc, float(conf_thres), 1 if multi_label else 0, 0, _p(pred), None, _p(cnt), None, None, None, _stream())
    incl = torch.cumsum(cnt.view(-1), 0, dtype=torch.int32)
    off = (incl - cnt.view(-1)).contiguous()
    per_image = cnt.sum(1, dtype=torch.int64)
    if class_ok is not None:
        per_image = torch.cat((per_image, class_ok.reshape(1).to(torch.int64)))
    per_image = per_image.tolist()                             # the one host read: sizes the outputs, and the caller's segment list
    ok = bool(per_image.pop()) if class_ok is not None else True
    total = sum(per_image)
    by_class = bool(segment_by_class) and ok and not (max_per_image is not None and per_image and max(per_image) > max_per_image)
    flags = (1 if multi_label else 0, 1 if by_class else 0)
    key = torch.empty((total,), dtype=torch.int64, device=pred.device)
    anchor = torch.empty((total,), dtype=torch.int32, device=pred.device)
    cls = torch.empty((total,), dtype=torch.int32, device=pred.device)
    if total:
        call("sy11_nms_candidates", B, D, A, nc, float(conf_thres), *flags, _p(pred), _p(off), None, _p(key), _p(anchor), _p(cls), _stream())
    return key, anchor, cls, per_image, by_class


def nms_sorted_segments(boxes_sorted: torch.Tensor, seg_of_row: torch.Tensor, nseg: int, iou_thres: float, max_keep: int) -> torch.Tensor:
    """Greedy NMS of ``nseg`` segments at once; ``seg_of_row`` (int64, non-decreasing) names the segment of every row of
    ``boxes_sorted`` ((n, 4) f32, a segment's rows in score-descending order).  -> bool keep mask (n,), at most ``max_keep``
    survivors per segment.  One host read (the longest segment and the workspace size)."""
    _need_gpu(boxes_sorted, seg_of_row)
    n = boxes_sorted.shape[0]
    keep = torch.zeros((n,), dtype=torch.uint8, device=boxes_sorted.device)
    if n == 0:
        return keep.bool()
    cnt = torch.bincount(seg_of_row, minlength=nseg)
    nw = (cnt + 63) // 64
    words = cnt * nw
    ws_end = torch.cumsum(words, 0)
    max_n, total_words = (int(v) for v in torch.stack((cnt.max(), ws_end[-1])).tolist())
    start = torch.zeros((nseg + 1,), dtype=torch.int32, device=boxes_sorted.device)
    start[1:] = torch.cumsum(cnt, 0)
    ws = (ws_end - words).contiguous()
    work = torch.empty((max(total_words, 1),), dtype=torch.int64, device=boxes_sorted.device)
    call("sy11_nms_sorted_segments", nseg, _p(start), _p(ws), max_n, _p(boxes_sorted.contiguous().float()), float(iou_thres), int(max_keep),
         _p(work), _p(keep), _stream())
    return keep.bool()


def stft_logmel(iq, window, mel_start, mel_w, n_fft, hop, n_frames, n_mel):
    """iq: (B, L) complex64 -> (db (B, frames, n_mel) f32, minmax (B, 2) f32)."""
    _need_gpu(iq, window, mel_start, mel_w)
    if iq.dtype != torch.complex64 or not iq.is_contiguous():
        raise _lib.Sy11Error("stft_logmel: iq must be contiguous complex64")
    B, L = iq.shape
    db = torch.empty((B, n_frames, n_mel), dtype=torch.float32, device=iq.device)
    mm = torch.empty((B, 2), dtype=torch.float32, device=iq.device)
    call("sy11_stft_minmax_init", B, _p(mm), _stream())
    call("sy11_stft_logmel", B, L, n_fft, hop, n_frames, n_mel, C.c_void_p(torch.view_as_real(iq).data_ptr()), _p(window),
         _p(mel_start), _p(mel_w), mel_w.shape[1], _p(db), _p(mm), _stream())
    return db, mm


def stft_normalize(db, mm, out=None):
    B, n_frames, n_mel = db.shape
    if out is not None and (tuple(out.shape) != (B, 3, n_mel, n_frames) or out.dtype != torch.float32 or not out.is_contiguous()
                            or out.device != db.device):
        raise _lib.Sy11Error("stft_normalize: `out` must be a contiguous f32 (B, 3, n_mel, n_frames) tensor on the same device")
    img = out if out is not None else torch.empty((B, 3, n_mel, n_frames), dtype=torch.float32, device=db.device)
    call("sy11_stft_normalize", B, n_mel, n_frames, _p(db), _p(mm), _p(img), _stream())
    return img


# ------------------------------------------------------------------------------------------------ image side of preprocess
def _img_dt(t):
    return _lib.U8 if t.dtype == torch.uint8 else _DT[t.dtype]


def image_u8_to_float(x, dtype=torch.float32, out=None):
    """batch["img"].float() / 255 (models/yolo/detect/train.py:59) in one pass; `out` may be a graph's static input."""
    if x.dtype != torch.uint8 or not x.is_contiguous():
        raise _lib.Sy11Error("image_u8_to_float: x must be a contiguous uint8 tensor")
    y = out if out is not None else torch.empty(x.shape, dtype=dtype, device=x.device)
    if tuple(y.shape) != tuple(x.shape) or not y.is_contiguous() or y.device != x.device:
        raise _lib.Sy11Error("image_u8_to_float: `out` must be a contiguous tensor of x's shape on x's device")
    call("sy11_image_u8_to_float", _DT[y.dtype], x.numel(), _p(x), _p(y), _stream())
    return y


def image_resize_bilinear(x, size, dtype=None, out=None):
    """F.interpolate(x, size=size, mode="bilinear", align_corners=False) on (B, C, H, W); a uint8 x is divided by 255
    first (the multi_scale branch of preprocess_batch, models/yolo/detect/train.py:60-73)."""
    if x.dim() != 4 or not x.is_contiguous():
        raise _lib.Sy11Error("image_resize_bilinear: x must be a contiguous (B, C, H, W) tensor")
    B, Cc, IH, IW = x.shape
    OH, OW = int(size[0]), int(size[1])
    dtype = dtype or (torch.float32 if x.dtype == torch.uint8 else x.dtype)
    y = out if out is not None else torch.empty((B, Cc, OH, OW), dtype=dtype, device=x.device)
    if tuple(y.shape) != (B, Cc, OH, OW) or not y.is_contiguous() or y.device != x.device:
        raise _lib.Sy11Error("image_resize_bilinear: `out` must be a contiguous (B, C, OH, OW) tensor on x's device")
    call("sy11_image_resize_bilinear", _img_dt(x), _DT[y.dtype], B * Cc, IH, IW, OH, OW, _p(x), _p(y), _stream())
    return y


def image_letterbox(src, dst, new_hw, top, left, fill=114, reverse_c=False, chw=True):
    """src (h, w, 3) uint8 HWC on the device -> dst: a (3, H, W) / (H, W, 3) uint8 or float tensor (one batch slot).
    LetterBox.__call__ (data/augment.py:1544-1591) + the BGR->RGB / HWC->CHW / /255 of predictor.preprocess."""
    if src.dtype != torch.uint8 or src.dim() != 3 or src.shape[2] != 3 or not src.is_contiguous():
        raise _lib.Sy11Error("image_letterbox: src must be a contiguous (h, w, 3) uint8 tensor")
    if dst.dim() != 3 or not dst.is_contiguous() or dst.device != src.device or dst.shape[0 if chw else 2] != 3:
        raise _lib.Sy11Error("image_letterbox: dst must be a contiguous (3, H, W) [chw] or (H, W, 3) tensor on src's device")
    H, W = (dst.shape[1], dst.shape[2]) if chw else (dst.shape[0], dst.shape[1])
    call("sy11_image_letterbox", _img_dt(dst), src.shape[0], src.shape[1], H, W, int(new_hw[0]), int(new_hw[1]), int(top),
         int(left), int(fill), int(bool(reverse_c)), int(bool(chw)), _p(src), _p(dst), _stream())
    return dst


def image_mosaic_warp(tiles, canvas_hw, dst, minv=None, hsv_lut=None, flip_ud=False, flip_lr=False, fill=114,
                      reverse_c=False, chw=True):
    """Render one augmented sample (sy11_image_mosaic_warp).  tiles = [(src (h, w, 3) uint8 device tensor, x1, y1, x2,
    y2, padw, padh)] (at most 4); minv = 6 floats of the inverted affine map or None; hsv_lut = (3, 256) uint8 numpy
    array or None; dst = (3, H, W) / (H, W, 3) uint8 or float device tensor."""
    import numpy as np
    n = len(tiles)
    if dst.dim() != 3 or not dst.is_contiguous() or dst.shape[0 if chw else 2] != 3:
        raise _lib.Sy11Error("image_mosaic_warp: dst must be a contiguous (3, H, W) [chw] or (H, W, 3) tensor")
    H, W = (dst.shape[1], dst.shape[2]) if chw else (dst.shape[0], dst.shape[1])
    srcs = (C.c_void_p * max(n, 1))()
    geom = (C.c_int32 * (8 * max(n, 1)))()
    for t, (src, x1, y1, x2, y2, padw, padh) in enumerate(tiles):
        if src.dtype != torch.uint8 or src.dim() != 3 or src.shape[2] != 3 or not src.is_contiguous() or src.device != dst.device:
            raise _lib.Sy11Error("image_mosaic_warp: every tile must be a contiguous (h, w, 3) uint8 tensor on dst's device")
        srcs[t] = src.data_ptr()
        geom[8 * t:8 * t + 8] = [src.shape[0], src.shape[1], int(x1), int(y1), int(x2), int(y2), int(padw), int(padh)]
    m = (C.c_double * 6)(*[float(v) for v in minv]) if minv is not None else None
    lut = None
    if hsv_lut is not None:
        lut_np = np.ascontiguousarray(hsv_lut, dtype=np.uint8)
        if lut_np.size != 768:
            raise _lib.Sy11Error("image_mosaic_warp: hsv_lut must hold 3 x 256 bytes")
        lut = lut_np.ctypes.data_as(C.c_void_p)
    call("sy11_image_mosaic_warp", _img_dt(dst), n, srcs, geom, int(canvas_hw[0]), int(canvas_hw[1]), m, H, W, lut,
         int(bool(flip_ud)), int(bool(flip_lr)), int(fill), int(bool(reverse_c)), int(bool(chw)), _p(dst), _stream())
    return dst


class DetLossWorkspace:
    """Device buffers of one fused-loss evaluation (kept for the backward launch)."""

    def __init__(self, maps, strides, nc, gt):
        self.maps = maps
        self.nl = len(maps)
        self.B = maps[0].shape[0]
        self.nc = nc
        self.G = gt.shape[1]
        self.gt = gt.contiguous().float()
        dev = maps[0].device
        self.A = sum(m.shape[1] * m.shape[2] for m in maps)
        self.ptrs = (C.c_void_p * self.nl)(*[m.data_ptr() for m in maps])
        self.hs = (C.c_int32 * self.nl)(*[m.shape[1] for m in maps])
        self.ws = (C.c_int32 * self.nl)(*[m.shape[2] for m in maps])
        self.st = (C.c_float * self.nl)(*[float(s) for s in strides])
        B, A, G = self.B, self.A, max(self.G, 1)
        self.pbox = torch.empty((B, A, 4), dtype=torch.float32, device=dev)
        self.align = torch.empty((B, G, A), dtype=torch.float32, device=dev)
        self.overlap = torch.empty((B, G, A), dtype=torch.float32, device=dev)
        self.topk = torch.empty((B, G, 10), dtype=torch.int32, device=dev)
        self.assign = torch.empty((B, A), dtype=torch.int32, device=dev)
        self.norm = torch.empty((B, A), dtype=torch.float32, device=dev)
        self.zero = torch.zeros(2 * B * G + 256, dtype=torch.float32, device=dev)      # pos maxima + sums, one memset
        self.pos = self.zero[: 2 * B * G]
        self.sums = self.zero[2 * B * G:].view(64, 4)


def det_loss_assign(maps, strides, nc, gt):
    """Stage 1 of the criterion (decode + task-aligned assignment + normalised alignment and its sum `tss`):
    maps: contiguous NHWC f32 (B,H,W,64+nc) list; gt (B,G,5).  -> workspace holding assign / norm / sums[:, 0]."""
    _need_gpu(*maps, gt)
    for m in maps:
        if m.dtype != torch.float32 or not m.is_contiguous() or m.shape[-1] != 64 + nc:
            raise _lib.Sy11Error("det_loss: maps must be contiguous NHWC f32 with 64+nc channels")
    w = DetLossWorkspace(maps, strides, nc, gt)
    cast = lambda a: C.cast(a, C.c_void_p)
    call("sy11_det_loss_assign", w.B, nc, w.nl, cast(w.ptrs), cast(w.hs), cast(w.ws), cast(w.st), w.G, _p(w.gt), _p(w.pbox),
         _p(w.align), _p(w.overlap), _p(w.topk), _p(w.assign), _p(w.pos), _p(w.norm), _p(w.sums), _stream())
    return w


def det_loss_terms(w: DetLossWorkspace):
    """Stage 2: box / cls / dfl partial sums into sums[:, 1:4] for the assignment (w.assign, w.norm) the workspace holds."""
    cast = lambda a: C.cast(a, C.c_void_p)
    call("sy11_det_loss_terms", w.B, w.nc, w.nl, cast(w.ptrs), cast(w.hs), cast(w.ws), cast(w.st), w.G, _p(w.gt), _p(w.assign),
         _p(w.norm), _p(w.sums), _stream())
    return w


def det_loss_forward(maps, strides, nc, gt):
    """maps: contiguous NHWC f32 (B,H,W,64+nc) list; gt (B,G,5).  -> workspace (sums (64,4) = [tss, box, cls, dfl] partials)."""
    return det_loss_terms(det_loss_assign(maps, strides, nc, gt))


def det_loss_finish(w: DetLossWorkspace, gains):
    """-> device tensor [loss, box, cls, dfl, 1 / max(tss, 1)] (loss.py:268-275), one launch."""
    out = torch.empty(5, dtype=torch.float32, device=w.sums.device)
    call("sy11_det_loss_finish", _p(w.sums), w.B, float(gains[0]), float(gains[1]), float(gains[2]), _p(out), _stream())
    return out


def det_loss_pack_targets(batch_idx, cls, bboxes, B, G, scale_wh):
    """v8DetectionLoss.preprocess: three (n, ...) f32 device columns -> (B, G, 5) [cls, xyxy pixels]; strided views are fine."""
    _need_gpu(batch_idx, cls, bboxes)
    n = batch_idx.shape[0]
    gt = torch.empty((B, G, 5), dtype=torch.float32, device=batch_idx.device)
    if n == 0 or G == 0:
        return gt.zero_() if G else gt
    for t in (batch_idx, cls, bboxes):
        if t.dtype != torch.float32:
            raise _lib.Sy11Error("det_loss_pack_targets: target columns must be f32")
    bi, cl = batch_idx.reshape(n, -1)[:, 0], cls.reshape(n, -1)[:, 0]
    if bboxes.dim() != 2 or bboxes.shape[1] != 4 or bboxes.stride(1) != 1:
        raise _lib.Sy11Error("det_loss_pack_targets: bboxes must be (n, 4) with unit inner stride")
    call("sy11_det_loss_pack_targets", n, B, G, _p(bi), max(bi.stride(0), 1), _p(cl), max(cl.stride(0), 1), _p(bboxes), max(bboxes.stride(0), 4),
         float(scale_wh[0]), float(scale_wh[1]), _p(gt), _stream())
    return gt


def det_loss_backward(w: DetLossWorkspace, upstream: torch.Tensor, gains, out=None, inv_tss=None):
    """d loss / d maps.  ``upstream``: device scalar (upstream gradient; already divided by max(tss, 1) unless ``inv_tss`` — the
    device scalar 1 / max(tss, 1) of det_loss_finish — is given).  ``out``: optional list of NHWC f32 buffers to write into (the
    captured backward graph's static output-gradient tensors) — used when every shape matches, else fresh tensors are returned."""
    if out is not None and len(out) == w.nl and all(o.shape == m.shape and o.dtype == m.dtype and o.is_contiguous() for o, m in zip(out, w.maps)):
        dmaps = list(out)
    else:
        dmaps = [torch.empty_like(m) for m in w.maps]
    dptrs = (C.c_void_p * w.nl)(*[d.data_ptr() for d in dmaps])
    cast = lambda a: C.cast(a, C.c_void_p)
    call("sy11_det_loss_bwd", w.B, w.nc, w.nl, cast(w.ptrs), cast(dptrs), cast(w.hs), cast(w.ws), cast(w.st), w.G, _p(w.gt),
         _p(w.assign), _p(w.norm), _p(upstream), _p(inv_tss), float(gains[0]), float(gains[1]), float(gains[2]), _stream())
    return dmaps


# ------------------------------------------------------------------------------------------------ Fusion('ESChannel')
def fusion_stats(x, mm, amax, sq_slice):
    """x: NHWC view (B,H,W,C); mm (B,H,W,2) f32; amax (B,H,W) int16/uint16; sq_slice: (B, C) view of the (B, n*C) buffer."""
    _need_gpu(x, mm, amax, sq_slice)
    B, H, W, Cn = x.shape
    call("sy11_fusion_stats", dt_code(x.dtype), B, H * W, Cn, _p(x), view_ld(x), _p(mm), _p(amax), _p(sq_slice), sq_slice.stride(0), _stream())


def sab_map_fwd(mm, w18, S):
    B, H, W, _ = mm.shape
    call("sy11_sab_map_fwd", B, H, W, _p(mm), _p(w18), _p(S), _stream())


def sab_map_bwd(dS, S, mm, w18, dmm, dw18):
    B, H, W, _ = mm.shape
    call("sy11_sab_map_bwd", B, H, W, _p(dS), _p(S), _p(mm), _p(w18), _p(dmm), _p(dw18), _stream())


def gct_gate_fwd(sq, alpha, gamma, beta, eps, G):
    B, Ct = sq.shape
    call("sy11_gct_gate_fwd", B, Ct, _p(sq), _p(alpha), _p(gamma), _p(beta), float(eps), _p(G), _stream())


def gct_gate_bwd(sq, alpha, gamma, beta, eps, dG, q, dalpha, dgamma, dbeta):
    B, Ct = sq.shape
    call("sy11_gct_gate_bwd", B, Ct, _p(sq), _p(alpha), _p(gamma), _p(beta), float(eps), _p(dG), _p(q), _p(dalpha), _p(dgamma),
         _p(dbeta), _stream())


def fusion_combine(xs, Ss, G, out):
    _need_gpu(*xs, out)
    B, H, W, Cn = xs[0].shape
    a = []
    for i in range(3):
        a += [_p(xs[i]), view_ld(xs[i]), _p(Ss[i])] if i < len(xs) else [None, 0, None]
    call("sy11_fusion_combine", dt_code(out.dtype), B, H * W, Cn, len(xs), *a, _p(G), _p(out), view_ld(out), _stream())
    return out


def fusion_bwd_reduce(dout, x, dG_slice, dS):
    B, H, W, Cn = x.shape
    call("sy11_fusion_bwd_reduce", dt_code(x.dtype), B, H * W, Cn, _p(dout), view_ld(dout), _p(x), view_ld(x), _p(dG_slice),
         dG_slice.stride(0), _p(dS), _stream())


def fusion_bwd_apply(dout, x, G_slice, q_slice, S, dmm, amax, dx, accumulate):
    B, H, W, Cn = x.shape
    call("sy11_fusion_bwd_apply", dt_code(x.dtype), B, H * W, Cn, _p(dout), view_ld(dout), _p(x), view_ld(x), _p(G_slice), _p(q_slice),
         G_slice.stride(0), _p(S), _p(dmm), _p(amax), _p(dx), view_ld(dx), int(accumulate), _stream())


# ------------------------------------------------------------------------------------------------ trainer step (flat buffers)
OPT_PARTS = 1024


def opt_workspace(device):
    return torch.zeros(_lib.load().sy11_opt_workspace_floats(OPT_PARTS), dtype=torch.float32, device=device)


def opt_step(param, grad, mom, sq, ema, buf, ema_buf, ws, group_end, lr, momentum, weight_decay, kind, ema_decay, max_norm=10.0,
             beta2=0.999, eps=1e-8, scale=None, growth_tracker=None, adam_step=None, growth=2.0, backoff=0.5, interval=2000,
             norm_out=None):
    """engine/trainer.py:585-593 in two launches (csrc/optim.hip): unscale + ||g||, then clip + SGD-nesterov / AdamW + EMA +
    zero_grad + GradScaler.update.  All tensors are flat f32 device buffers; ``scale`` / ``growth_tracker``: GradScaler's device
    scalars (None = no AMP); ``ema`` / ``ema_buf`` None = no EMA."""
    _need_gpu(param, grad, mom, sq, ema, buf, ema_buf, ws)
    n = param.numel()
    for t in (grad, mom, sq, ema):
        if t is not None and (t.numel() != n or t.dtype != torch.float32 or not t.is_contiguous()):
            raise _lib.Sy11Error("opt_step: grad / mom / sq / ema must be contiguous f32 buffers of the parameter buffer's length")
    d = OptDesc()
    d.n, d.n_buf = n, (buf.numel() if (buf is not None and ema_buf is not None) else 0)
    for k in range(3):
        d.group_end[k], d.lr[k], d.momentum[k], d.weight_decay[k] = int(group_end[k]), float(lr[k]), float(momentum[k]), float(weight_decay[k])
    d.kind, d.beta2, d.eps, d.max_norm, d.ema_decay = int(kind), float(beta2), float(eps), float(max_norm), float(ema_decay)
    d.amp = int(scale is not None)
    d.growth_factor, d.backoff_factor, d.growth_interval, d.nparts = float(growth), float(backoff), int(interval), OPT_PARTS
    call("sy11_opt_grad_norm", n, _p(grad), _p(scale), _p(adam_step), _p(ws), OPT_PARTS, _stream())
    call("sy11_opt_step", C.byref(d), _p(param), _p(grad), _p(mom), _p(sq), _p(ema), _p(buf) if d.n_buf else None,
         _p(ema_buf) if d.n_buf else None, _p(ws), _p(scale), _p(growth_tracker), _p(adam_step), _p(norm_out), _stream())
