"""Training step with the reference trainer's semantics (ultralytics/engine/trainer.py: _setup_train :230-316,
_do_train :318-474, optimizer_step :585-593, build_optimizer :758-819), stripped to the hot path:
AMP (fp16 operands + GradScaler), nbs=64 accumulation, 3 parameter groups, grad-clip 10, EMA, RCCL gradient sum.
Dataset / callbacks / checkpoints / validation loops are outside the hot path (out of scope)."""
from __future__ import annotations

import math
import random
from types import SimpleNamespace

import torch
import torch.nn as nn

from ..utils.torch_utils import ModelEMA
from . import ddp

DEFAULTS = dict(lr0=0.01, momentum=0.937, weight_decay=0.0005, nbs=64, box=7.5, cls=0.5, dfl=1.5, amp=True,
                optimizer="SGD", warmup_epochs=3.0, warmup_momentum=0.8, warmup_bias_lr=0.1, multi_scale=False, imgsz=640,
                deterministic=None)      # None: leave libsy11's option as it is (SY11_DETERMINISTIC); True / False: cfg/default.yaml:29


class DetectionTrainer:
    """One-process-per-GPU trainer for a ``DetectionModel`` (the reference's DetectionTrainer hot path)."""

    def __init__(self, model, batch_size=64, device="cuda", overrides=None, world_size=1, producer=None, graphs=True,
                 flat=True):
        self.args = SimpleNamespace(**{**DEFAULTS, **(overrides or {})})
        self.device = torch.device(device)
        self.model = model.to(self.device)
        self.model.args = self.args                       # v8DetectionLoss reads box / cls / dfl gains here
        self.batch_size = batch_size
        self.world_size = world_size
        self.data_parallel = world_size > 1 or ddp.REHEARSE     # issue the collectives (world size 1 only in the one-rank rehearsal)
        self.producer = producer                           # optional IQ -> image producer (SpectrogramProducer)
        self.amp = bool(self.args.amp)
        if self.args.deterministic is not None and self.device.type == "cuda":
            from ..utils.torch_utils import set_deterministic
            set_deterministic(bool(self.args.deterministic))
        self.model._sy11_dtype = torch.float16 if self.amp else torch.float32
        for k, v in self.model.named_parameters():          # trainer.py:246-252: always freeze DFL
            if ".dfl" in k:
                v.requires_grad = False
        self.scaler = torch.amp.GradScaler("cuda", enabled=self.amp)
        self.accumulate = max(round(self.args.nbs / (batch_size * world_size)), 1)
        wd = self.args.weight_decay * batch_size * world_size * self.accumulate / self.args.nbs
        if self.args.optimizer == "auto":                   # trainer.py:778-786: the default of the reference's cfg
            nc = getattr(self.model, "nc", None) or self.model.model[-1].nc
            iterations = getattr(self.args, "iterations", 100000)
            name, lr, mom = ("SGD", 0.01, 0.9) if iterations > 10000 else ("AdamW", round(0.002 * 5 / (4 + nc), 6), 0.9)
            self.args.optimizer, self.args.lr0, self.args.momentum, self.args.warmup_bias_lr = name, lr, mom, 0.0
        self.flat = None
        if flat:
            # parameters / buffers re-homed into flat buffers; 3 optimizer groups = 3 slices (see engine/flat.py)
            from . import GradStore
            from .flat import FlatEMA, FlatState
            self.flat = FlatState(self.model)
            store = GradStore(self.model, order=self.flat.order)
            store.external_zero = True
            store.begin_backward(self.device)
            self.model.__dict__["_sy11_grads"] = store
            self.model.__dict__["_sy11_flat"] = self.flat
            if self.data_parallel:                           # DDP's constructor: rank-0 weights and buffers to everybody (two
                ddp.broadcast_parameters(self.model)         # broadcasts of the flat buffers), before the EMA takes its copy
            self.grad_store = store
            self.flat_params = [t.requires_grad_(True) for t in self.flat.group_tensors(self.flat.flat)]
            self.flat_grads = self.flat.group_tensors(store.flat)
            self.optimizer = self.build_flat_optimizer(self.flat_params, self.args.optimizer, self.args.lr0,
                                                       self.args.momentum, wd)
            self.ema = FlatEMA(self.model, self.flat)
            self._hip_step_init()
        else:
            if self.data_parallel:
                ddp.broadcast_parameters(self.model)
            self.optimizer = self.build_optimizer(self.model, self.args.optimizer, self.args.lr0, self.args.momentum, wd)
            self.ema = ModelEMA(self.model)
        if self.data_parallel:
            ddp.attach(self.model)
            # identical kernels on every rank: only rank 0 measures tile configurations (first eager step); its picks are
            # broadcast after the eager warm-up steps, before the graphs are captured (train_step)
            import os
            self.rank = int(os.environ.get("RANK", "0"))
            if self.device.type == "cuda" and self.rank != 0:
                from .. import _lib
                _lib.set_option("tune", 0)
        if graphs and not self.args.multi_scale:            # hipGraph replay of forward/backward after 2 eager steps (multi_scale draws a new
            # input size every step: one captured graph + activation pool per size would not pay, those runs stay eager)
            from . import enable_graphs
            enable_graphs(self.model)
        self.last_opt_step = -1
        self.ni = 0
        self._sig_runs = {}                                 # input signature -> training steps seen (tuner-pick sharing, train_step)

    @staticmethod
    def build_optimizer(model, name="SGD", lr=0.01, momentum=0.9, decay=1e-5):
        """Three groups by name/type (trainer.py:776-813): biases (no decay), norm weights (no decay), the rest."""
        g = [], [], []
        bn = tuple(v for k, v in nn.__dict__.items() if "Norm" in k)
        for module_name, module in model.named_modules():
            for param_name, param in module.named_parameters(recurse=False):
                if not param.requires_grad:
                    continue
                fullname = f"{module_name}.{param_name}" if module_name else param_name
                if "bias" in fullname:
                    g[2].append(param)
                elif isinstance(module, bn):
                    g[1].append(param)
                else:
                    g[0].append(param)
        if name in {"Adam", "Adamax", "AdamW", "NAdam", "RAdam"}:
            opt = getattr(torch.optim, name)(g[2], lr=lr, betas=(momentum, 0.999), weight_decay=0.0)
        elif name == "SGD":
            opt = torch.optim.SGD(g[2], lr=lr, momentum=momentum, nesterov=True)
        else:
            raise NotImplementedError(f"optimizer {name}")
        opt.add_param_group({"params": g[0], "weight_decay": decay})
        opt.add_param_group({"params": g[1], "weight_decay": 0.0})
        return opt

    @staticmethod
    def build_flat_optimizer(flat_params, name="SGD", lr=0.01, momentum=0.9, decay=1e-5):
        """Same three groups as build_optimizer, each ONE flat tensor: [decay weights, norm weights, biases]."""
        w, n, b = flat_params
        # fused=True: ONE kernel per flat group, and — decisive under AMP — GradScaler.step() hands found_inf / grad_scale to
        # the kernel instead of reading found_inf back on the host.  The reference's per-tensor optimizer makes that
        # `.item()` every step (engine/trainer.py:590); here it stalled the launch queue for ~3 ms of a 29 ms step (r01 trace).
        fused = {"fused": True} if b.is_cuda else {}
        if name in {"Adam", "AdamW"}:
            opt = getattr(torch.optim, name)([b], lr=lr, betas=(momentum, 0.999), weight_decay=0.0, **fused)
        elif name in {"Adamax", "NAdam", "RAdam"}:
            opt = getattr(torch.optim, name)([b], lr=lr, betas=(momentum, 0.999), weight_decay=0.0)
        elif name == "SGD":
            opt = torch.optim.SGD([b], lr=lr, momentum=momentum, nesterov=True, **fused)
        else:
            raise NotImplementedError(f"optimizer {name}")
        opt.add_param_group({"params": [w], "weight_decay": decay})
        opt.add_param_group({"params": [n], "weight_decay": 0.0})
        if fused and name == "SGD":
            # the fused kernel returns early on found_inf, but torch has by then allocated the momentum buffers with
            # torch.empty() and will treat them as valid from the next step on: start from explicit zeros instead
            # (momentum * 0 + g == the reference's first-step buffer = clone(g))
            for t in (b, w, n):
                opt.state[t]["momentum_buffer"] = torch.zeros_like(t)
        return opt

    # ---- optimizer step as two HIP launches (csrc/optim.hip)
    _FLAT_TO_GROUP = (1, 2, 0)          # flat slices are [decay, norm, bias]; the torch optimizer's groups (the reference's order) [bias, decay, norm]

    def _hip_step_init(self):
        """On the MI355X the flat trainer steps through sy11_opt_grad_norm + sy11_opt_step; the torch optimizer object stays the
        holder of the hyper-parameters (warm-up writes lr / momentum there) and of the STATE: its momentum buffers / Adam moments
        become views of flat buffers this method owns, so state_dict() / checkpoints are unchanged."""
        self._hip_step = self.flat is not None and self.device.type == "cuda" and type(self.optimizer).__name__ in ("SGD", "Adam", "AdamW")
        if not self._hip_step:
            return
        from .. import ops as K
        n = self.flat.flat.numel()
        adam = type(self.optimizer).__name__ != "SGD"
        # 0 SGD-nesterov, 1 AdamW (decoupled decay), 2 Adam (torch.optim.Adam: the weight group's decay is L2 ADDED TO THE GRADIENT
        # before the moments — the reference's build_optimizer gives 'Adam' exactly that; r03 ran it through the AdamW rule)
        self._opt_kind = {"SGD": 0, "AdamW": 1, "Adam": 2}[type(self.optimizer).__name__]
        self._opt_mom = torch.zeros(n, dtype=torch.float32, device=self.device)
        self._opt_sq = torch.zeros(n, dtype=torch.float32, device=self.device) if adam else None
        self._adam_step = torch.zeros((), dtype=torch.float32, device=self.device) if adam else None
        self._opt_ws = K.opt_workspace(self.device)
        if self.amp:
            self.scaler._lazy_init_scale_growth_tracker(self.device)      # the two device scalars the step kernel updates in place
        self._adopt_optimizer_state()

    def _adopt_optimizer_state(self):
        """(Re-)home the torch optimizer's state tensors into the flat momentum / moment buffers (after construction and after
        load_state_dict, which replaces them with copies)."""
        if not getattr(self, "_hip_step", False):
            return
        keys = ("exp_avg", "exp_avg_sq") if self._opt_kind else ("momentum_buffer",)
        bufs = (self._opt_mom, self._opt_sq) if self._opt_kind else (self._opt_mom,)
        for (a, b), t in zip(self.flat.group_slices, self.flat_params):
            st = self.optimizer.state[t]
            for key, flat in zip(keys, bufs):
                view = flat[a:b]
                old = st.get(key)
                if old is not None and old.data_ptr() != view.data_ptr():
                    view.copy_(old.to(view.device, torch.float32).reshape(-1))
                st[key] = view
            if self._opt_kind:
                old = st.get("step")
                if old is not None and old is not self._adam_step:
                    self._adam_step.fill_(float(old))
                st["step"] = self._adam_step

    def _hip_optimizer_step(self):
        from .. import ops as K
        gs = self.optimizer.param_groups
        order = self._FLAT_TO_GROUP
        lr = [gs[j]["lr"] for j in order]
        mom = [(gs[j]["betas"][0] if self._opt_kind else gs[j]["momentum"]) for j in order]
        wd = [gs[j]["weight_decay"] for j in order]
        ema_on = bool(self.ema) and getattr(self.ema, "enabled", True)
        d = 0.0
        if ema_on:
            self.ema.updates += 1
            d = self.ema.decay(self.ema.updates)
        amp = self.amp and self.scaler.is_enabled()
        K.opt_step(self.flat.flat, self.grad_store.flat, self._opt_mom, self._opt_sq,
                   self.ema.ema_state.flat if ema_on else None, self.flat.flat_buf if ema_on else None,
                   self.ema.ema_state.flat_buf if ema_on else None, self._opt_ws, [b for _, b in self.flat.group_slices], lr, mom, wd,
                   self._opt_kind, d, max_norm=10.0,
                   beta2=gs[0]["betas"][1] if self._opt_kind else 0.999, eps=gs[0].get("eps", 1e-8) if self._opt_kind else 1e-8,
                   scale=self.scaler._scale if amp else None, growth_tracker=self.scaler._growth_tracker if amp else None,
                   adam_step=self._adam_step, growth=self.scaler.get_growth_factor() if amp else 2.0,
                   backoff=self.scaler.get_backoff_factor() if amp else 0.5, interval=self.scaler.get_growth_interval() if amp else 2000)

    # ---- the epoch loop (engine/trainer.py:318-474 `_do_train`, without callbacks / plots / logging)
    def fit(self, train_loader, epochs, val_batches=None, save_dir=None, close_mosaic=10, start_epoch=0, lrf=0.01, cos_lr=False,
            patience=100):
        """Epochs of: [close mosaic for the last `close_mosaic` epochs] -> iterations (warm-up, accumulate, step, EMA) ->
        scheduler step -> validate the EMA model -> fitness -> save last / best.  ``val_batches``: a callable returning an
        iterable of validation batches (or None: no validation).  Returns the per-epoch records.  Every rank runs the same
        number of iterations; validation and checkpoints happen on rank 0 only (RANK env), as in the reference."""
        import os
        from pathlib import Path
        from .checkpoint import save_checkpoint
        from .validator import DetectionValidator
        rank0 = int(os.environ.get("RANK", "-1")) in (-1, 0)
        self.set_schedule(len(train_loader), epochs, lrf=lrf, cos_lr=cos_lr)
        self.epoch = start_epoch
        if start_epoch:
            for g in self.optimizer.param_groups:
                g["lr"] = g["initial_lr"] * self.lf(start_epoch)
            self.ni = start_epoch * self.nb
            self.last_opt_step = self.ni - 1
        # resume_training() restores the best fitness so far (trainer.py:727-745): a worse first epoch must not replace best.pt
        best, stale, history = float(getattr(self, "best_fitness", None) or -1.0), 0, []
        mosaic_closed = False
        save_dir = Path(save_dir) if save_dir else None
        if save_dir and rank0:
            save_dir.mkdir(parents=True, exist_ok=True)
        for epoch in range(start_epoch, epochs):
            if not mosaic_closed and epoch >= epochs - close_mosaic and hasattr(train_loader.dataset, "close_mosaic"):
                train_loader.dataset.close_mosaic(train_loader.dataset.hyp)       # trainer.py:341-343; `>=` + latch: a run resumed
                mosaic_closed = True                                              # past that epoch closes it at once (trainer.py:752-756)
            if hasattr(train_loader, "set_epoch"):
                train_loader.set_epoch(epoch)
            tloss = None
            for i, batch in enumerate(train_loader):
                _, items = self.train_step(batch)
                tloss = items if tloss is None else (tloss * i + items) / (i + 1)  # trainer.py:385-387 running mean of the 3 loss items
            self.end_epoch()
            rec = {"epoch": epoch, "train_loss": [float(v) for v in tloss] if tloss is not None else None, "metrics": None, "fitness": None}
            if rank0 and val_batches is not None:
                metrics = DetectionValidator(self.ema.ema if self.ema else self.model, device=self.device,
                                             half=self.amp)(self.ema.ema if self.ema else self.model, val_batches())
                rec["metrics"] = metrics
                rec["fitness"] = float(metrics.get("fitness", 0.1 * metrics.get("metrics/mAP50(B)", 0.0) + 0.9 * metrics.get("metrics/mAP50-95(B)", 0.0)))
            if rank0 and save_dir:
                save_checkpoint(save_dir / "last.pt", trainer=self, epoch=epoch, best_fitness=max(best, rec["fitness"] or -1.0),
                                train_metrics=rec["metrics"])
                if rec["fitness"] is None or rec["fitness"] >= best:
                    save_checkpoint(save_dir / "best.pt", trainer=self, epoch=epoch, best_fitness=rec["fitness"], train_metrics=rec["metrics"])
            if rec["fitness"] is not None:
                stale = 0 if rec["fitness"] >= best else stale + 1               # EarlyStopping (utils/torch_utils.py:713-757)
                best = max(best, rec["fitness"])
            history.append(rec)
            self.best_fitness = best
            stop = bool(patience and stale >= patience)
            if self.data_parallel and torch.distributed.is_available() and torch.distributed.is_initialized():
                # only rank 0 validates, so only rank 0 knows: every rank must leave the loop together, or the others hang in
                # the next epoch's gradient all-reduce (trainer.py:456-461 broadcasts the same flag)
                flag = [stop]
                torch.distributed.broadcast_object_list(flag, 0)
                stop = bool(flag[0])
            if stop:
                break
        return history

    # ---- resume (engine/trainer.py:728-756)
    def resume_training(self, ckpt):
        """Take optimizer state, EMA and epoch counter from a checkpoint dict (this build's or the reference's).  The model
        weights are the caller's business, as in the reference (the model is built from the checkpoint first).  Returns the
        epoch to continue with."""
        start_epoch = ckpt.get("epoch", -1) + 1
        if ckpt.get("optimizer") is not None:
            self._load_optimizer_state(ckpt["optimizer"])
        if self.ema and ckpt.get("ema") is not None:
            self.ema.ema.load_state_dict(ckpt["ema"].float().state_dict())      # in place: the flat EMA buffers keep their views
            self.ema.updates = ckpt.get("updates", 0)
        self.epoch = start_epoch
        bf = ckpt.get("best_fitness")
        self.best_fitness = float(bf) if bf is not None else None          # fit() seeds its `best` from this
        return start_epoch

    def _load_optimizer_state(self, osd):
        """An optimizer state_dict in this trainer's flat layout (one tensor per group) loads as it is; the reference's
        per-parameter layout — groups [biases, decayed weights, norm weights], parameters in named_modules order — is
        scattered into the flat momentum / Adam buffers slice by slice."""
        groups = osd["param_groups"]
        flat_layout = all(len(g["params"]) == 1 for g in groups)
        if self.flat is None or flat_layout:
            self.optimizer.load_state_dict(osd)
            self._adopt_optimizer_state()
            return
        from .flat import _view_like, param_groups
        mine = param_groups(self.model)                                # (decay, norm, bias) lists, reference iteration order
        lists = (mine[2], mine[0], mine[1])                            # the reference's group order: bias, decay, norm
        if len(groups) != 3 or any(len(g["params"]) != len(l) for g, l in zip(groups, lists)):
            raise ValueError("optimizer state does not match this model's parameter groups "
                             f"({[len(g['params']) for g in groups]} vs {[len(l) for l in lists]})")
        starts = {0: self.flat.group_slices[2][0], 1: self.flat.group_slices[0][0], 2: self.flat.group_slices[1][0]}
        for k, (g, plist) in enumerate(zip(groups, lists)):
            og = self.optimizer.param_groups[k]
            flat_t = og["params"][0]
            st = self.optimizer.state[flat_t]
            for key, val in g.items():
                if key != "params" and key in og and key not in ("fused", "foreach", "capturable", "differentiable", "maximize"):
                    og[key] = val
            for idx, p in zip(g["params"], plist):
                ps = osd["state"].get(idx)
                if not ps:
                    continue
                off = self.flat.offsets[id(p)] - starts[k]
                for name, val in ps.items():
                    if torch.is_tensor(val) and val.dim() > 0 and val.numel() == p.numel():
                        if name not in st:
                            st[name] = torch.zeros_like(flat_t)
                        _view_like(st[name][off:off + p.numel()], p).copy_(val.to(flat_t.device, torch.float32))
                    elif name == "step":
                        st["step"] = torch.as_tensor(float(val), dtype=torch.float32, device=flat_t.device if self.optimizer.defaults.get("fused") else "cpu")
        self._adopt_optimizer_state()

    # ---- learning-rate schedule and warm-up (engine/trainer.py:209-215, :330, :364-377, :430-433)
    def set_schedule(self, batches_per_epoch, epochs, lrf=0.01, cos_lr=False):
        """Arm the per-iteration warm-up and the per-epoch LambdaLR of the reference for a run of ``epochs`` x ``batches_per_epoch``."""
        import math
        self.nb, self.epochs, self.epoch = int(batches_per_epoch), int(epochs), 0
        if cos_lr:
            self.lf = lambda x: max((1 - math.cos(x * math.pi / self.epochs)) / 2, 0) * (lrf - 1) + 1      # one_cycle(1, lrf, epochs)
        else:
            self.lf = lambda x: max(1 - x / self.epochs, 0) * (1.0 - lrf) + lrf
        for g in self.optimizer.param_groups:
            g.setdefault("initial_lr", g["lr"])
            g["lr"] = g["initial_lr"] * self.lf(0)
        self.nw = max(round(self.args.warmup_epochs * self.nb), 100) if self.args.warmup_epochs > 0 else -1
        self.last_opt_step = -1
        self.ni = 0

    def _warmup(self):
        """Per-iteration warm-up: bias lr falls from warmup_bias_lr, the other lrs rise from 0, momentum rises from
        warmup_momentum, the accumulation window grows from 1 to nbs / batch_size."""
        ni = self.ni
        if getattr(self, "nw", -1) < 0 or ni > self.nw:
            return
        import numpy as np
        xi = [0, self.nw]
        self.accumulate = max(1, int(np.interp(ni, xi, [1, self.args.nbs / (self.batch_size * self.world_size)]).round()))
        for j, g in enumerate(self.optimizer.param_groups):
            g["lr"] = float(np.interp(ni, xi, [self.args.warmup_bias_lr if j == 0 else 0.0, g["initial_lr"] * self.lf(self.epoch)]))
            if "momentum" in g:
                g["momentum"] = float(np.interp(ni, xi, [self.args.warmup_momentum, self.args.momentum]))

    def end_epoch(self):
        """scheduler.step() of the reference: lr = initial_lr * lf(epoch) for the next epoch."""
        self.epoch += 1
        for g in self.optimizer.param_groups:
            g["lr"] = g["initial_lr"] * self.lf(self.epoch)

    def preprocess_batch(self, batch):
        """detect/train.py:57-74: uint8 -> float/255; with a producer: raw IQ -> spectrogram image on device."""
        if "iq" in batch and self.producer is not None:
            from . import graph_static_input
            iq = batch["iq"].to(self.device, non_blocking=True)
            shape = (iq.shape[0], 3, self.producer.n_mel, self.producer.n_frames)
            batch["img"] = self.producer(iq, out=graph_static_input(self.model, shape) if self.model.training else None)
        else:
            from . import graph_static_input
            from .. import ops as K
            img = batch["img"].to(self.device, non_blocking=True)
            size = tuple(img.shape[2:])
            if self.args.multi_scale:                          # train.py:60-73: the size draw, then one resize launch
                stride = int(max(self.model.stride))
                sz = random.randrange(int(self.args.imgsz * 0.5), int(self.args.imgsz * 1.5 + stride)) // stride * stride
                sf = sz / max(size)
                if sf != 1:
                    size = tuple(math.ceil(x * sf / stride) * stride for x in size)
            static = graph_static_input(self.model, (img.shape[0], img.shape[1], *size)) if self.model.training else None
            if size != tuple(img.shape[2:]):
                batch["img"] = K.image_resize_bilinear(img.contiguous(), size, dtype=torch.float32, out=static)
            elif img.dtype == torch.uint8:
                batch["img"] = K.image_u8_to_float(img.contiguous(), torch.float32, out=static)
            else:
                batch["img"] = img.float()
        return batch

    def batch_buffer(self, imgsz=None):
        """A ``n -> tensor | None`` for ``build_dataloader(out=...)``: the static input of the captured training graph
        for n images (None until the graph exists, then the loader renders every sample straight into it)."""
        from . import graph_static_input
        size = (imgsz, imgsz) if isinstance(imgsz, int) else tuple(imgsz or (self.args.imgsz, self.args.imgsz))
        return lambda n: graph_static_input(self.model, (n, 3, *size)) if self.model.training else None

    def optimizer_step(self):
        """trainer.py:585-593."""
        if getattr(self, "_hip_step", False):
            self._hip_optimizer_step()                   # unscale + norm ; clip + step + EMA + zero_grad + scale update: 2 launches
            return
        if self.flat is not None:
            for p, g in zip(self.flat_params, self.flat_grads):
                p.grad = g                               # flat slices of the GradStore buffer
            self.scaler.unscale_(self.optimizer)
            torch.nn.utils.clip_grad_norm_(self.flat_params, max_norm=10.0)      # same elements => same total norm
            self.scaler.step(self.optimizer)
            self.scaler.update()
            self.grad_store.flat.zero_()                 # one memset instead of zero_grad over 255 tensors
        else:
            self.scaler.unscale_(self.optimizer)
            torch.nn.utils.clip_grad_norm_(self.model.parameters(), max_norm=10.0)
            self.scaler.step(self.optimizer)
            self.scaler.update()
            self.optimizer.zero_grad()
        if self.ema:
            self.ema.update(self.model)

    def train_step(self, batch):
        """One iteration of the hot loop (trainer.py:378-393): preprocess, forward+loss, scaled backward
        (+ RCCL gradient sum), optimizer step every ``accumulate`` iterations."""
        self.model.train()
        self._warmup()
        batch = self.preprocess_batch(batch)
        loss, items = self.model(batch)
        will_step = self.ni - self.last_opt_step >= self.accumulate
        store = self.model.__dict__.get("_sy11_grads")
        if store is not None:
            store.defer_allreduce = not will_step                # data parallel: one gradient all-reduce per optimizer step
        self.scaler.scale(loss).backward()
        if will_step:                                            # trainer.py:391 (ni counts from 0, last_opt_step from -1)
            self.optimizer_step()
            self.last_opt_step = self.ni
        self.ni += 1
        if self.data_parallel and self.device.type == "cuda":
            # Rank 0 alone measures tile configurations, and it measures per PROBLEM, on the first eager call that meets it:
            # the first two steps of every input signature (a new image size, the short last batch of an epoch, the first
            # steps after a resume) run eagerly and may add picks, the third is captured.  Whether THIS rank just ran such a
            # step is a rank-local fact (multi_scale draws a size per rank; last batches can differ), so the decision to
            # broadcast is itself agreed on first: one MAX all-reduce of the flag on the host-side control group per step
            # (ddp.share_tuner_picks_if_any) — all ranks share, or none does.  Ranks >= 1 never measure (`tune` 0 by design).
            img = batch["img"]
            sig = (tuple(img.shape), img.dtype, self.model._sy11_dtype)
            n = self._sig_runs.get(sig, 0) + 1
            self._sig_runs[sig] = n
            ddp.share_tuner_picks_if_any(n <= 2)                 # collective, every step, on every rank
        return loss.detach(), items
