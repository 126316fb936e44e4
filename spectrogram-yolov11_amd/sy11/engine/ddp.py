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


# One-rank rehearsal (SY11_DDP_REHEARSE=1): every collective of the data-parallel path — parameter broadcast, tuner-pick broadcast,
# the control group's flag, the flat-gradient all-reduce in both flavours — goes through the backend with world size 1 instead of being
# skipped.  On a one-GPU box that is the only way RCCL itself (communicator set-up, its stream beside the replayed graphs) ever runs;
# a sum over one rank is the identity, so the run must reproduce the single-process trainer bit for bit (tests/test_ddp_nccl_gpu.py).
REHEARSE = os.environ.get("SY11_DDP_REHEARSE", "0") != "0"


def active(group=None) -> bool:
    """True when the collectives are to be issued: a process group of more than one rank (or the one-rank rehearsal)."""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or REHEARSE)


def setup_process_group(backend: str | None = None):
    """init_process_group from torchrun's env (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*); 'nccl' IS RCCL on ROCm."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # rehearsing N ranks on a box with fewer GPUs: SY11_FORCE_DEVICE puts every rank on that device, SY11_DDP_BACKEND=gloo carries the
    # collectives (RCCL refuses two ranks on one GPU) — bench.py --gpus N and YOLO(...).train(device=[...]) both come through here
    if "SY11_FORCE_DEVICE" in os.environ:
        local = int(os.environ["SY11_FORCE_DEVICE"])
    if backend is None:
        backend = os.environ.get("SY11_DDP_BACKEND") or None
    if (world > 1 or REHEARSE) and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def allreduce_flat(flat: torch.Tensor, group=None):
    """In-place SUM all-reduce of a flat gradient buffer (any backend)."""
    if active(group):
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat


# The two-bucket overlap below has run over gloo with two ranks on one GPU (tests/test_ddp_gpu.py: the collective does not run on a
# device stream of its own there) and over RCCL with ONE rank (r04, the rehearsal above: RCCL's stream beside the second backward
# graph, bit-identical to the single-process trainer) — a >= 2-GPU RCCL run of tests/test_ddp_nccl_gpu.py's two-rank test has still not
# been recorded, so it stays opt-in (SY11_DDP_OVERLAP=1); the default is ONE all-reduce of the flat buffer after backward: 37.8 MB
# over xGMI ~ 0.3 ms of a ~19 ms step.
OVERLAP = os.environ.get("SY11_DDP_OVERLAP", "0") != "0"
BUCKET_LAYER = 5      # layers >= 5 of the yolo11 graph hold > 99 % of the parameters; layers 0-4 (320^2 ... 80^2 maps) a third of the backward time


class _StagedAllReduce:
    """The gradient exchange of one model as two buckets over the flat gradient buffer (the reference's DDP buckets,
    engine/trainer.py:273, re-thought for one flat buffer):

      stage 0  fires when the backward pass has finished every layer >= ``bucket_layer``: their parameters' ranges of the flat
               buffer (a few contiguous runs: the tail of each optimizer group) are all-reduced ASYNCHRONOUSLY — RCCL runs them
               on its own stream beside the backward of the high-resolution layers that is still to come;
      stage 1  fires at the end of backward: the remaining few small runs are packed into one staging tensor, reduced with
               ONE call (latency bound, ~tens of KB), unpacked, and the stage-0 handles are waited for (stream-side waits).

    Without a stage-0 call (module without a bucket mark, accumulation window still open, eager sub-module) stage 1 reduces the
    whole flat buffer with one call — the r01 behaviour.  SUM semantics: DDP's mean x the reference's `loss *= world_size`."""

    staged = True

    def __init__(self, model, store, group=None, bucket_layer=BUCKET_LAYER):
        self.model, self.store, self.group, self.bucket_layer = model, store, group, bucket_layer
        self.late = self.early = None
        self.works = []
        self.stage0_done = False
        self.staging = None
        self.timing = None            # a list: (start, end) event pairs on the launch stream around each stage's collectives (bench.py)

    def _ranges(self):
        """(late, early) lists of [start, end) element ranges of store.flat, merged over adjacent parameters."""
        if self.late is not None:
            return
        names = {id(p): k for k, p in self.model.named_parameters()}

        def layer_of(p):
            parts = names.get(id(p), "").split(".")
            return int(parts[1]) if len(parts) > 2 and parts[0] == "model" and parts[1].isdigit() else 1 << 30
        late, early = [], []
        pad = lambda n: (n + 7) // 8 * 8                         # noqa: E731  (the GradStore / FlatState padding rule)
        for p in self.store.params:
            a = self.store.offsets[id(p)]
            b = a + pad(p.numel())
            dst = late if layer_of(p) >= self.bucket_layer else early
            if dst and dst[-1][1] == a:
                dst[-1][1] = b
            else:
                dst.append([a, b])
        self.late, self.early = late, early

    def __call__(self, s, stage):
        if self.timing is None:
            return self._run(s, stage)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        self._run(s, stage)
        e1.record()
        self.timing.append((e0, e1))

    def _run(self, s, stage):
        if getattr(s, "defer_allreduce", False):
            return
        if stage == 0:
            self.works, self.stage0_done = [], False                  # a backward that raised between the stages leaves nothing behind
            if not active(self.group):
                return
            self._ranges()
            if not self.late or not self.early:
                return
            self.works = [dist.all_reduce(s.flat[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True) for a, b in self.late]
            self.stage0_done = True
            return
        if not self.stage0_done:                                  # nothing was started early: the whole buffer, one call
            allreduce_flat(s.flat, self.group)
            return
        n = sum(b - a for a, b in self.early)
        if self.staging is None or self.staging.numel() != n or self.staging.device != s.flat.device:
            self.staging = torch.empty(n, dtype=s.flat.dtype, device=s.flat.device)
        views = [s.flat[a:b] for a, b in self.early]
        torch.cat(views, out=self.staging)
        dist.all_reduce(self.staging, op=dist.ReduceOp.SUM, group=self.group)
        off = 0
        for v in views:
            v.copy_(self.staging[off:off + v.numel()])
            off += v.numel()
        for w in self.works:
            w.wait()                                              # the current stream waits for the collective (no host block on RCCL)
        self.works, self.stage0_done = [], False


def attach(model: torch.nn.Module, group=None, bucket_layer=None):
    """Make every engine backward of ``model`` end with the gradient sum over ranks (no-op for world_size 1), the first
    bucket overlapped with the rest of backward when the model is a layer graph (BaseModel)."""
    store = model.__dict__.get("_sy11_grads")
    if store is None:
        store = GradStore(model)
        model.__dict__["_sy11_grads"] = store
    # gradient accumulation (accumulate > 1): the flat buffer sums the micro-steps, so it must be reduced ONCE, after the
    # last backward of the window (the trainer raises ``store.defer_allreduce`` on the others) — reducing it after every
    # backward would re-sum the already reduced part.  SUM is linear: allreduce(sum_k g_k) == sum_k allreduce(g_k).
    if bucket_layer is None:
        bucket_layer = BUCKET_LAYER if OVERLAP else 0
    layers = getattr(model, "model", None)
    if bucket_layer and layers is not None and hasattr(layers, "__len__") and len(layers) > bucket_layer + 1:
        model.__dict__["_sy11_bucket_layer"] = int(bucket_layer)
    module_post_backward[id(store)] = _StagedAllReduce(model, store, group, bucket_layer)
    return model


def share_tuner_picks(src: int = 0, group=None):
    """Every rank runs the kernels rank ``src`` measured: the tile autotuner's pick tables go from ``src`` to everybody
    (libsy11 sy11_tune_export / sy11_tune_import).  Collective: call on all ranks.  Returns the number of picks."""
    from .. import _lib
    if not active(group):
        return len(_lib.tune_export()) // 16
    box = [_lib.tune_export() if dist.get_rank(group) == src else None]
    dist.broadcast_object_list(box, src, group=group)
    if dist.get_rank(group) != src:
        _lib.load().sy11_tune_clear()
        _lib.tune_import(box[0])
    return len(box[0]) // 16


_CTL_GROUP = None


def control_group():
    """A host-side (gloo) group next to the RCCL one, for the few per-step CONTROL decisions every rank must take identically.
    A collective on it blocks only the host thread — which runs a step ahead of the GPU — never the launch stream, and needs no
    device synchronisation to read its result.  Collective: created on first use by all ranks together."""
    global _CTL_GROUP
    if not active():
        return None
    if _CTL_GROUP is None:
        _CTL_GROUP = dist.group.WORLD if dist.get_backend() == "gloo" else dist.new_group(backend="gloo")
    return _CTL_GROUP


def share_tuner_picks_if_any(local_flag: bool, src: int = 0) -> bool:
    """Rank-uniform trigger for share_tuner_picks().  Every rank calls this EVERY training step with its own flag ("this step ran
    eagerly and may have added picks"); the flags are MAX-reduced on the control group, so either all ranks enter the broadcast
    or none does.  (r03 decided from a rank-local counter keyed on the input signature: with `multi_scale`, where every rank
    draws its own image size — random.randrange under seed + 1 + RANK — or an uneven last batch, one rank called the broadcast
    while another went on to the gradient all-reduce: mismatched collectives.)"""
    g = control_group()
    if g is None:
        return False
    t = torch.tensor([1 if local_flag else 0], dtype=torch.int32)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=g)
    if int(t.item()) == 0:
        return False
    share_tuner_picks(src)
    return True


def free_port() -> int:
    """A free TCP port on the loopback interface (utils/dist.py:13-22 find_free_network_port)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch(script_args, nproc: int, env=None, timeout=None):
    """One process per GPU on this node, as the reference's `generate_ddp_command` + subprocess.run do for
    `YOLO(...).train(device=[0, 1, ...])` (utils/dist.py:25-66, engine/trainer.py:170-207): RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_ADDR / MASTER_PORT in the environment of ``nproc`` CHILD processes running ``python <script_args...>``.
    Children are spawned, never exec'ed over this process, and the caller must not have initialised the GPU in a way the
    children inherit (they are fresh interpreters).  Returns the list of exit codes; raises if any is non-zero."""
    import subprocess
    import sys
    port = free_port()
    base = dict(os.environ if env is None else env)
    base.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(nproc), HSA_ENABLE_IPC_MODE_LEGACY=base.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = []
    for r in range(nproc):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, *script_args], env=e))
    codes = []
    try:
        for p in procs:
            codes.append(p.wait(timeout=timeout))
    finally:
        for p in procs:                                  # never leave ranks behind (exact PIDs, no pattern kills)
            if p.poll() is None:
                p.kill()
    if any(codes):
        raise RuntimeError(f"data-parallel launch failed: exit codes {codes}")
    return codes


def broadcast_parameters(model: torch.nn.Module, src: int = 0, group=None):
    """Rank-0 weights and buffers to everybody (what DDP's constructor does).  With a flat layout (engine/flat.py) that is two
    broadcasts of the flat buffers plus the few tensors living outside them; otherwise one per tensor (through a contiguous
    staging copy when a backend cannot take the tensor's strides)."""
    if not active(group):
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
