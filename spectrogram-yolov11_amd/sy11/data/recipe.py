"""Pixel-free samples for loader WORKER PROCESSES (the reference's DataLoader workers, data/build.py:129-157).

The reference's workers each run the whole CPU pipeline (decode, mosaic, warpAffine, HSV, flips) and ship a finished 1.2 MB image per
sample to the training process.  Here the pixels live on the MI355X, so a worker process — which must never touch the GPU — runs only
the part that is host work anyway: the RANDOM DRAWS and the LABEL GEOMETRY of the transforms (data/augment.py), on images that exist
as ``LazyImage`` nodes: a shape plus how to obtain the pixels later (a source file, a resize of another node, a rendered recipe).
What crosses the process boundary per sample is ~1.5 KB — tile rectangles, one inverted affine map, three 256-byte colour tables, two
flip bits, the boxes — and the training process turns it into pixels with one fused launch per sample (sy11_image_mosaic_warp) after
resolving the source nodes against its cache of decoded images in HBM.
"""
from __future__ import annotations

import torch


class LazyImage:
    """Stand-in for a (h, w, 3) uint8 device tensor inside a worker process.  ``op`` is one of
         ("file", dataset index)                                    the decoded source image
         ("letterbox", node, (new_h, new_w), top, left, fill)       sy11_image_letterbox of another node into this node's shape
         ("render", recipe state)                                   a DeviceImage recipe flattened to pixels (a second warp, ...)
    Carries exactly the tensor surface the transforms look at (shape / dim / dtype / is_cuda / contiguous); no arithmetic."""

    __slots__ = ("shape", "op")
    dtype = torch.uint8
    is_cuda = False
    device = torch.device("cpu")

    def __init__(self, hw, op):
        self.shape = (int(hw[0]), int(hw[1]), 3)
        self.op = op

    def dim(self):
        return 3

    def contiguous(self):
        return self

    def __reduce__(self):
        return (LazyImage, (self.shape[:2], self.op))

    def key(self):
        """Hashable identity of the pixels this node stands for (cache key in the training process); None for one-off renders."""
        if self.op[0] == "file":
            return ("file", self.op[1])
        if self.op[0] == "letterbox":
            inner = self.op[1].key()
            return None if inner is None else ("letterbox", inner, self.shape[:2], *self.op[2:])
        return None


def file_image(index, hw):
    return LazyImage(hw, ("file", int(index)))


def letterbox_image(src, out_hw, new_hw, top, left, fill=114):
    return LazyImage(out_hw, ("letterbox", src, (int(new_hw[0]), int(new_hw[1])), int(top), int(left), int(fill)))


def is_lazy(x):
    return isinstance(x, LazyImage)


class Materializer:
    """Training-process side: LazyImage -> uint8 tensor on the device.  ``fetch(index)`` returns the decoded source image of a
    dataset index as a device tensor; resized / letterboxed nodes are kept in a small LRU so that the mosaic partners a worker keeps
    re-drawing from its buffer are produced once."""

    def __init__(self, fetch, device, capacity=4096, max_bytes=None):
        self.fetch, self.device, self.capacity = fetch, torch.device(device), capacity
        # bounded by BYTES as well as by entries: the cache holds raw decoded sources of any resolution next to their
        # letterboxed copies; the reference bounds its buffer to min(n, 8 x batch, 1000) already-resized images (data/base.py:98-104).
        # Default budget: an eighth of the device memory that is free when the loader is built, at most 8 GiB.
        if max_bytes is None:
            max_bytes = 2 << 30
            if self.device.type == "cuda":
                try:
                    max_bytes = min(torch.cuda.mem_get_info(self.device)[0] // 8, 8 << 30)
                except Exception:                             # noqa: BLE001 — no usable device query: keep the fixed budget
                    pass
        self.max_bytes, self.bytes = int(max_bytes), 0
        self.cache = {}

    def __call__(self, node):
        if not is_lazy(node):
            return node
        from .. import ops as K
        key = node.key()
        hit = self.cache.get(key) if key is not None else None
        if hit is not None:
            return hit
        kind = node.op[0]
        if kind == "file":
            out = self.fetch(node.op[1])
        elif kind == "letterbox":
            _, src, new_hw, top, left, fill = node.op
            out = torch.empty(node.shape, dtype=torch.uint8, device=self.device)
            K.image_letterbox(self(src), out, new_hw, top, left, fill, reverse_c=False, chw=False)
        elif kind == "render":
            out = node.op[1].resolved(self).render(chw=False)
        else:
            raise ValueError(f"unknown lazy image op {kind!r}")
        if key is not None:
            size = out.numel() * out.element_size()
            if len(self.cache) >= self.capacity or self.bytes + size > self.max_bytes:
                # insertion order = age: drop the oldest quarter of the entries, and keep dropping until the newcomer fits
                drop = max(len(self.cache) // 4, 1)
                for k in list(self.cache):
                    if drop <= 0 and self.bytes + size <= self.max_bytes:
                        break
                    v = self.cache.pop(k)
                    self.bytes -= v.numel() * v.element_size()
                    drop -= 1
            if size <= self.max_bytes:
                self.cache[key] = out
                self.bytes += size
        return out
