"""Checkpoint interchange with the reference (ultralytics/engine/trainer.py:512-543 save_model; nn/tasks.py:820-960
torch_safe_load / attempt_load_one_weight).

A reference ``.pt`` is a pickled dict whose ``ema`` / ``model`` entries are pickled *module objects*: class paths under
``ultralytics.*`` plus each module's ``__dict__`` (``_parameters``, ``_buffers``, ``_modules``, plain attributes).  The
sy11 modules keep the reference's attribute names and state_dict keys, so such an object is re-created here by
resolving every ``ultralytics.*`` class path to its sy11 counterpart while unpickling — no ultralytics import needed.
Saving writes the same dictionary keys; the EMA is pickled as sy11 modules (loadable by this package) and its
``state_dict`` is stored beside it so that a reference installation can rebuild the model from ``yaml`` + weights.
"""
from __future__ import annotations

import io
import pickle
import types
from copy import deepcopy
from datetime import datetime

import torch
import torch.nn as nn

__all__ = ("load_checkpoint", "attempt_load_one_weight", "save_checkpoint", "checkpoint_dict")


class IterableSimpleNamespace(types.SimpleNamespace):
    """Stand-in for ultralytics.utils.IterableSimpleNamespace (the pickled ``model.args``)."""

    def __iter__(self):
        return iter(vars(self).items())

    def get(self, key, default=None):
        return getattr(self, key, default)


class _InertHolder(nn.Module):
    """Unpickling target for reference classes whose behaviour lives inside the fused HIP criterion here."""

    def forward(self, *a, **k):
        from .. import _lib
        raise _lib.Sy11Error(f"{type(self).__name__} is a checkpoint placeholder: the criterion runs as sy11.utils.loss.v8DetectionLoss")


class BboxLoss(_InertHolder):
    """ultralytics.utils.loss.BboxLoss placeholder (module level: instances stay picklable)."""


class DFLoss(_InertHolder):
    """ultralytics.utils.loss.DFLoss placeholder."""


class TaskAlignedAssigner(_InertHolder):
    """ultralytics.utils.tal.TaskAlignedAssigner placeholder."""


def _class_table():
    from ..nn import tasks
    from ..nn.modules import block, conv, head
    from ..utils import loss
    t = {("ultralytics.nn.tasks", n): getattr(tasks, n) for n in ("DetectionModel", "BaseModel")}
    for mod, names in ((conv, ("Conv", "DWConv", "DDWConv", "Concat", "Fusion", "GCT", "WeightedSpatialAttention")),
                       (block, ("DFL", "SPPF", "C2f", "C3", "C3k", "C3k2", "Bottleneck", "Attention", "PSABlock", "C2PSA")),
                       (head, ("Detect",))):
        for n in names:
            t[("ultralytics.nn.modules." + mod.__name__.rsplit(".", 1)[1], n)] = getattr(mod, n)
            t[("ultralytics.nn.modules", n)] = getattr(mod, n)
    t[("ultralytics.utils", "IterableSimpleNamespace")] = IterableSimpleNamespace
    # a reference model pickled after its first loss call carries `criterion` (v8DetectionLoss with its BboxLoss / DFLoss /
    # TaskAlignedAssigner children).  None of that state is used here — the criterion is rebuilt by init_criterion() — so the
    # children unpickle into the inert holders above and `_adopt` drops `criterion` from every loaded model.
    t[("ultralytics.utils.loss", "v8DetectionLoss")] = loss.v8DetectionLoss
    t[("ultralytics.utils.loss", "BboxLoss")] = BboxLoss
    t[("ultralytics.utils.loss", "DFLoss")] = DFLoss
    t[("ultralytics.utils.tal", "TaskAlignedAssigner")] = TaskAlignedAssigner
    return t


class _RefUnpickler(pickle.Unpickler):
    table = None

    def find_class(self, module, name):
        if module.startswith("ultralytics"):
            if _RefUnpickler.table is None:
                _RefUnpickler.table = _class_table()
            cls = _RefUnpickler.table.get((module, name))
            if cls is None:
                from .. import _lib
                raise _lib.Sy11Error(f"checkpoint references {module}.{name}, which is outside the MI355X hot path (no sy11 counterpart)")
            return cls
        return super().find_class(module, name)


_ref_pickle = types.ModuleType("sy11_ref_pickle")
_ref_pickle.Unpickler = _RefUnpickler
_ref_pickle.load = lambda f, **kw: _RefUnpickler(f, **kw).load()
_ref_pickle.__name__ = "pickle"


def _adopt(model: nn.Module) -> nn.Module:
    """Make an unpickled reference module tree a valid sy11 tree: f32 master weights, filters in channels_last memory,
    the per-filter group tag the engine keeps, no stale engine caches."""
    from ..nn.modules.conv import Conv
    model = model.float()
    # a criterion pickled by the reference carries the reference's attributes, not this build's (stride_f, _gains ...):
    # drop it, BaseModel.loss rebuilds it on first use
    model.__dict__.pop("criterion", None)
    for m in model.modules():
        for k in [k for k in m.__dict__ if k.startswith("_sy11_")]:
            del m.__dict__[k]
        if isinstance(m, Conv):
            w = m.conv.weight
            w.data = w.data.contiguous(memory_format=torch.channels_last)
            w._sy11_groups = m.conv.groups
    return model


def load_checkpoint(path_or_file, map_location="cpu"):
    """torch_safe_load: -> the checkpoint dict, with every ``ultralytics.*`` object re-created from sy11 classes."""
    if hasattr(path_or_file, "read"):
        return torch.load(path_or_file, map_location=map_location, pickle_module=_ref_pickle, weights_only=False)
    with open(path_or_file, "rb") as f:
        return torch.load(f, map_location=map_location, pickle_module=_ref_pickle, weights_only=False)


def attempt_load_one_weight(path_or_file, device=None, inplace=True, fuse=False):
    """nn/tasks.py:937-960: (model, ckpt) — the EMA if present else ``model``, in eval mode, args merged from train_args."""
    ckpt = load_checkpoint(path_or_file)
    args = {**(ckpt.get("train_args") or {})}
    model = _adopt(ckpt.get("ema") or ckpt["model"])
    model.args = IterableSimpleNamespace(**{k: v for k, v in args.items() if k in ("box", "cls", "dfl", "imgsz", "data", "task", "single_cls")}) \
        if args else getattr(model, "args", None)
    model.pt_path = str(path_or_file) if not hasattr(path_or_file, "read") else None
    if not hasattr(model, "stride"):
        model.stride = torch.tensor([32.0])
    if device is not None:
        model = model.to(device)
    model = (model.fuse() if fuse and hasattr(model, "fuse") else model).eval()
    for m in model.modules():
        if hasattr(m, "inplace"):
            m.inplace = inplace
    return model, ckpt


def _optimizer_state_fp16(sd):
    """convert_optimizer_state_dict_to_fp16 (utils/torch_utils.py): f32 state tensors -> f16, 'step' left alone."""
    for state in sd.get("state", {}).values():
        for k, v in state.items():
            if k != "step" and isinstance(v, torch.Tensor) and v.dtype is torch.float32:
                state[k] = v.half()
    return sd


def checkpoint_dict(ema_model: nn.Module, updates=0, optimizer=None, epoch=0, best_fitness=None, train_args=None, train_metrics=None,
                    train_results=None):
    """The dictionary trainer.save_model writes (same keys); ``ema`` = detached f16 copy of the EMA model."""
    # engine caches (captured graphs, flat state, grad store) never travel — and a CUDAGraph cannot be deep-copied: set them
    # aside for the copy
    stash = []
    for m in ema_model.modules():
        held = {k: m.__dict__.pop(k) for k in [k for k in m.__dict__ if k.startswith("_sy11_")]}
        if held:
            stash.append((m, held))
    try:
        ema = deepcopy(ema_model)
    finally:
        for m, held in stash:
            m.__dict__.update(held)
    for p in ema.parameters():                               # parameters may be views of a flat buffer: give them own storage
        p.data = p.data.clone()
    for b in ema.buffers():
        b.data = b.data.clone()
    ema = ema.cpu().half()
    return {
        "epoch": epoch, "best_fitness": best_fitness, "model": None, "ema": ema, "updates": updates,
        "optimizer": _optimizer_state_fp16(deepcopy(optimizer.state_dict())) if optimizer is not None else None,
        "train_args": dict(train_args or {}), "train_metrics": dict(train_metrics or {}), "train_results": train_results or {},
        "date": datetime.now().isoformat(), "version": "sy11-r01", "license": "AGPL-3.0 (https://ultralytics.com/license)",
        "docs": "https://docs.ultralytics.com",
        # interchange for a reference installation (its unpickler cannot resolve sy11 classes): yaml + reference-keyed weights
        "sy11_yaml": getattr(ema_model, "yaml", None), "sy11_state_dict": {k: v.clone() for k, v in ema.state_dict().items()},
    }


def save_checkpoint(path, trainer=None, **kw):
    """save_model: serialise once, write ``path``.  ``trainer``: a sy11 DetectionTrainer (EMA, optimizer, args taken from it)."""
    if trainer is not None:
        kw = {"ema_model": trainer.ema.ema, "updates": trainer.ema.updates, "optimizer": trainer.optimizer,
              "train_args": vars(trainer.args), **kw}
    buf = io.BytesIO()
    torch.save(checkpoint_dict(**kw), buf)
    data = buf.getvalue()
    if hasattr(path, "write"):
        path.write(data)
    else:
        with open(path, "wb") as f:
            f.write(data)
    return len(data)
