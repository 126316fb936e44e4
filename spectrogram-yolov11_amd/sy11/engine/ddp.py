"""Single-node data parallelism: one process per GPU, gradients summed by ONE RCCL all-reduce over the flat f32
gradient buffer (37.8 MB for yolo11s) instead of DDP's per-bucket hooks (engine/trainer.py:217-228, :273 in the
reference).  xGMI is point-to-point, so a single large message lets RCCL drive all 7 links of the mesh.

Semantics kept: the reference wraps in DDP (mean over ranks) and multiplies the loss by world_size
(trainer.py:381-382), i.e. the applied gradient is the SUM of the per-rank gradients — exactly what a SUM
all-reduce of the un-scaled per-rank gradients gives.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

from . import GradStore, module_post_backward


def setup_process_group(backend: str | None = None):
    """init_process_group from torchrun's env (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*); 'nccl' IS RCCL on ROCm."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def allreduce_flat(flat: torch.Tensor, group=None):
    """In-place SUM all-reduce of a flat gradient buffer (any backend)."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


def attach(model: torch.nn.Module, group=None):
    """Make every engine backward of ``model`` end with the gradient all-reduce (no-op for world_size 1)."""
    store = model.__dict__.get("_sy11_grads")
    if store is None:
        store = GradStore(model)
        model.__dict__["_sy11_grads"] = store
    # gradient accumulation (accumulate > 1): the flat buffer sums the micro-steps, so it must be all-reduced ONCE, after the
    # last backward of the window (the trainer raises ``store.defer_allreduce`` on the others) — reducing it after every
    # backward would re-sum the already reduced part.  SUM is linear: allreduce(sum_k g_k) == sum_k allreduce(g_k).
    def hook(s):
        if not getattr(s, "defer_allreduce", False):
            allreduce_flat(s.flat, group)
    module_post_backward[id(store)] = hook
    return model


def broadcast_parameters(model: torch.nn.Module, src: int = 0, group=None):
    """Rank-0 weights and buffers to everybody (what DDP's constructor does).  With a flat layout (engine/flat.py) that is two
    broadcasts of the flat buffers plus the few tensors living outside them; otherwise one per tensor (through a contiguous
    staging copy when a backend cannot take the tensor's strides)."""
    if not (dist.is_initialized() and dist.get_world_size(group) > 1):
        return
    flat = model.__dict__.get("_sy11_flat")
    covered = []
    if flat is not None:
        for buf in (flat.flat, flat.flat_buf):
            if buf.numel():
                dist.broadcast(buf, src, group=group)
                covered.append((buf.data_ptr(), buf.data_ptr() + buf.numel() * buf.element_size()))
    for t in list(model.parameters()) + list(model.buffers()):
        d = t.data
        if any(lo <= d.data_ptr() < hi for lo, hi in covered):
            continue
        if d.is_contiguous():
            dist.broadcast(d, src, group=group)
        else:
            c = d.contiguous()
            dist.broadcast(c, src, group=group)
            d.copy_(c)
