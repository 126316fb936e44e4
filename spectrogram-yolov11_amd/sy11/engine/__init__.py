"""Execution engine: NHWC activations, an explicit backward tape and the autograd bridge.

Why not torch autograd per op?  The hot path writes producers straight into channel slices of concat buffers,
accumulates gradients in place into slices, and keeps raw conv outputs (not BN/SiLU outputs) for backward.  An
explicit tape gives that control; ONE ``torch.autograd.Function`` (``EngineFn``) exposes a whole module (or the whole
DetectionModel) to torch so ``loss.backward()``, ``GradScaler`` and optimizers work unchanged
(engine/trainer.py:378-393 surface).

Parameter gradients are written by the kernels directly into views of ONE flat f32 buffer (``GradStore``), which is
what the data-parallel step all-reduces (one RCCL call over xGMI instead of per-tensor buckets).
"""
from __future__ import annotations

import os

from typing import List, Optional

import torch

from .. import ops


class Act:
    """An NHWC activation view plus its (lazily allocated) gradient.  Children are channel slices of a root buffer."""

    __slots__ = ("_data", "root", "c0", "_grad", "_ready", "raw", "req", "_dtype")

    def __init__(self, data: Optional[torch.Tensor], root: Optional["Act"] = None, c0: int = 0,
                 raw: Optional[torch.Tensor] = None, req: bool = True, dtype=None):
        self._data = data
        self.root = root
        self.c0 = c0
        self._grad = None
        self._ready = False
        self.raw = raw          # original NCHW f32 image (the stem kernel reads it directly; NHWC copy made lazily)
        self.req = req          # does anything upstream want this activation's gradient?
        self._dtype = dtype

    @property
    def data(self) -> torch.Tensor:
        if self._data is None:
            t = self.raw
            v = torch.empty((t.shape[0], t.shape[2], t.shape[3], t.shape[1]), dtype=self._dtype, device=t.device)
            v.copy_(t.permute(0, 2, 3, 1))
            self._data = v
        return self._data

    @property
    def shape(self):
        if self._data is None:
            t = self.raw
            return torch.Size((t.shape[0], t.shape[2], t.shape[3], t.shape[1]))
        return self._data.shape

    @property
    def C(self):
        return self.shape[-1]

    def slice(self, a: int, b: int) -> "Act":
        r = self.root or self
        return Act(self.data[..., a:b], r, self.c0 + a, req=self.req)

    # ---- gradients
    def _root_grad(self):
        r = self.root or self
        if r._grad is None:
            r._grad = torch.empty(r.data.shape, dtype=r.data.dtype, device=r.data.device)
        return r

    def grad_for_write(self):
        """-> (grad view, accumulate?).  First full-buffer writer overwrites; everything else accumulates."""
        r = self._root_grad()
        if r._ready:
            acc = True
        elif self.root is None:
            acc = False
        else:                       # a slice is written before the whole buffer: start from zeros
            r._grad.zero_()
            acc = True
        r._ready = True
        g = r._grad if self.root is None else r._grad[..., self.c0:self.c0 + self.C]
        return g, acc

    def grad_read(self):
        r = self._root_grad()
        if not r._ready:
            r._grad.zero_()
            r._ready = True
        return r._grad if self.root is None else r._grad[..., self.c0:self.c0 + self.C]

    def set_grad(self, g: torch.Tensor):
        assert self.root is None
        self._grad = g
        self._ready = True


# Filter gradients on a second stream, SY11_WGRAD_STREAM of them per fork (0 = everything on one stream).  Inside the captured
# graph a cross-stream edge costs ~10 us, so one fork per layer LOSES (r01: 26.55 -> 27.05 ms; r03: 19.93 -> 20.65) — but the
# replayed graph otherwise runs strictly one kernel at a time (rocprofv3 trace of the r03 build: 643 kernels, 0.000 ms with two
# kernels in flight), and the filter gradients (MFMA / LDS bound, 2.6 TB/s) pair well with the BatchNorm passes (HBM bound) of
# the layers below them.  Batched, the fork count drops to three per step: same-box sweep 0 / 16 / 24 / 32 / 48 / 64 / 100 per
# fork = 19.73 / 19.48 / 19.36 / 19.26 / 19.45 / 19.47 / 19.76 ms (32 again 19.26, 0 again 19.61; second box 19.94 -> 19.55).
_SIDE_WGRAD = os.environ.get("SY11_WGRAD_STREAM", "32") != "0"
# how many filter-gradient launches share one fork (one cross-stream edge per batch instead of one per layer)
_SIDE_BATCH = max(int(os.environ.get("SY11_WGRAD_STREAM", "32") or 0), 1)
# (Measured and dropped: forking by WORK instead of count — a batch per 0.5 / 1 / 2 / 4 / 8 x 1e8 gradient elements, so that the
# few large layers at the end of the backward pass overlap too: 20.23 / 19.76 / 19.89 / 19.50 / 19.61 ms against 19.52-19.56 by
# count.  The large maps' filter gradients are HBM-bound like the BatchNorm passes they would run beside; the pairing pays on the
# small maps, where neither kernel fills the chip.  An EARLIER first fork — 12 / 20 / 8 launches, then 32 / 40 / 36 — also loses:
# 19.72 / 19.74 / 20.06 ms against 19.57-19.59.)
_SIDE_STREAMS = {}


def _side_stream(device):
    d = torch.device(device)
    key = d.index if d.index is not None else torch.cuda.current_device()
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=d)
    return _SIDE_STREAMS[key]


# Independent sub-graphs on their own streams (the three levels of the Detect head): inside the captured graph they become
# parallel branches, so the small kernels of the 40x40 / 20x20 levels (200-800 workgroups on 256 CUs) could run beside the 80x80
# level's instead of after it.  OFF by default: measured r02 (two A/B pairs in one process sequence on one MI355X) 23.11 / 22.88 ms
# per step with the branches against 22.40 / 22.38 ms on one stream — as with the filter-gradient stream above, parallel branches
# of a hipGraph cost more than the idle CUs they fill.  SY11_BRANCH_STREAMS=1 turns it on.
_BRANCH_STREAMS_ON = os.environ.get("SY11_BRANCH_STREAMS", "0") != "0"
# (r03 also carried SY11_HEAD_HOIST=1 — the Detect chains of the stride-8 / 16 levels launched as soon as their feature map existed, beside
# the neck layers that follow it: -0.9 % in a same-box A/B, but it broke _Branch's contract below (main-stream launches inside an open
# branch section) and the fusion variant's f16 parity test failed with it on (model.15's filter / BN-weight gradients 6 % / 11 % off).
# r04 removed it: a 1 % lever with a known wrong answer is not worth carrying; multi-problem launches are the replacement.)
_BRANCH_STREAMS = {}


def _branch_stream(device, b):
    d = torch.device(device)
    key = (d.index if d.index is not None else torch.cuda.current_device(), b)
    if key not in _BRANCH_STREAMS:
        _BRANCH_STREAMS[key] = torch.cuda.Stream(device=d)
    return _BRANCH_STREAMS[key]


class _Branch:
    """``with ec.branch(b):`` — the enclosed launches (and the tape closures recorded inside) run on branch stream ``b``, ordered
    after everything the current stream has issued so far.  The caller ends the group with ``ec.join_branches()`` and issues
    nothing on the main stream in between (that is what makes cross-stream buffer reuse safe: every branch section starts with a
    wait on the main stream, every consumer on the main stream comes after the join)."""

    def __init__(self, ec, b):
        self.ec, self.b, self.cm = ec, b, None

    def __enter__(self):
        ec = self.ec
        self.start = len(ec.tape)
        if ec.use_branches:
            st = _branch_stream(ec.device, self.b)
            ec.main_stream = torch.cuda.current_stream(ec.device)
            st.wait_stream(ec.main_stream)
            self.cm = torch.cuda.stream(st)
            self.cm.__enter__()
            ec.open_branches.add(self.b)
        return self

    def __exit__(self, *exc):
        ec = self.ec
        if self.cm is not None:
            self.cm.__exit__(*exc)
        if ec.use_branches and len(ec.tape) > self.start:
            ec.tape_branches.append((self.start, len(ec.tape), self.b))
        return False


class Ctx:
    """State of one forward pass through the engine."""

    def __init__(self, training: bool, record: bool, dtype: torch.dtype, device, grads: Optional["GradStore"],
                 pool_hint: int = 0):
        self.training = training
        self.record = record          # build a backward tape
        self.dtype = dtype
        self.device = device
        self.tape: List = []
        self.marks = {}               # name -> tape length at a point of the forward pass ("bucket": first closure of the split layer)
        self.grads = grads
        # one memset for all the small f32 accumulators (BN statistics, backward sums) of a step instead of ~160 fills
        self.pool = torch.zeros(pool_hint, dtype=torch.float32, device=device) if pool_hint > 0 else None
        self.pool_off = 0
        self.pool_need = 0
        self.bn_counters: List[torch.Tensor] = []
        self.capturing = False        # inside hipGraph capture: no host-side caching keyed on parameter versions
        self.w_views = None           # {id(param): working-dtype [O][KH][KW][I] view} from ONE flat cast (trainer's FlatState)
        self.wt_views = None          # {id(param): tap-transposed view}, built lazily at the start of backward
        self.w_bank = None
        self.flat_state = None
        # filter gradients run on a second stream: wgrad(L) only needs dy(L) and the saved input, so it overlaps the
        # dgrad / BN-backward chain of the layers below it (each kernel alone leaves CUs idle in its prologue / tail)
        self.side = None
        self.side_refs: List = []
        self.use_side = _SIDE_WGRAD and device is not None and torch.device(device).type == "cuda"
        self.use_branches = _BRANCH_STREAMS_ON and device is not None and torch.device(device).type == "cuda"
        self.open_branches = set()
        self.main_stream = None           # the stream a branch section was entered from (set while any branch is open)
        self.tape_branches: List = []     # (first tape index, end, branch) of the closures recorded inside a branch section

    def on_side(self, fn, *keep):
        """Run ``fn`` (kernel launches) on the side stream, ordered after everything issued so far on the current stream.
        ``keep``: tensors the side work reads — held until join_side() so the allocator cannot recycle them early."""
        if not self.use_side:
            fn()
            return
        if self.side is None:
            self.side = _side_stream(self.device)
            self.side_queue = []
        self.side_queue.append(fn)
        self.side_refs.append(keep)
        if len(self.side_queue) >= _SIDE_BATCH:
            self.flush_side()

    def flush_side(self):
        """Fork: everything queued so far runs on the side stream, ordered after what EVERY stream of this pass has issued: the
        current one, the main stream when the flush happens inside a branch closure, and every branch still open (the
        32nd launch of a batch may be queued inside a branch closure — the batch also holds launches whose operands were produced on
        the main stream or on a sibling branch, and waiting for the current stream alone does not order those)."""
        if self.side is None or not self.side_queue:
            return
        cur = torch.cuda.current_stream(self.device)
        self.side.wait_stream(cur)
        if self.main_stream is not None and self.main_stream != cur:
            self.side.wait_stream(self.main_stream)
        for b in sorted(self.open_branches):
            st = _branch_stream(self.device, b)
            if st != cur:
                self.side.wait_stream(st)
        with torch.cuda.stream(self.side):
            for fn in self.side_queue:
                fn()
        self.side_queue = []

    def branch(self, b):
        return _Branch(self, b)

    def join_branches(self):
        """The current stream waits for every branch opened since the last join."""
        if self.open_branches:
            cur = torch.cuda.current_stream(self.device)
            for b in sorted(self.open_branches):
                cur.wait_stream(_branch_stream(self.device, b))
            self.open_branches.clear()

    def run_tape(self, hi, lo, at=None, then=None):
        """Backward closures hi, hi-1, ..., lo.  Closures recorded inside a branch section run on that branch's stream again
        (entered with a wait on the main stream, which has issued nothing since the section's far end); the first main-stream
        closure after them — and the end of the range — joins.  ``then()`` runs right after closure ``at`` (a main-stream index)."""
        spans = self.tape_branches
        for idx in range(hi, lo - 1, -1):
            b = None
            for (a0, a1, bb) in spans:
                if a0 <= idx < a1:
                    b = bb
                    break
            if b is None:
                self.join_branches()
                self.tape[idx]()
            else:
                st = _branch_stream(self.device, b)
                if b not in self.open_branches:
                    self.main_stream = torch.cuda.current_stream(self.device)
                    st.wait_stream(self.main_stream)
                    self.open_branches.add(b)
                with torch.cuda.stream(st):
                    self.tape[idx]()
            if idx == at and then is not None:
                self.join_branches()
                then()
        self.join_branches()

    def join_side(self):
        if self.side is not None and self.side_refs:
            self.flush_side()
            torch.cuda.current_stream(self.device).wait_stream(self.side)
            self.side_refs.clear()

    def attach_flat(self, module):
        """Trainer-provided FlatState: derive all working-dtype filters with one cast (and later one transpose) launch."""
        fs = module.__dict__.get("_sy11_flat")
        if fs is not None and self.training:
            self.flat_state = fs
            self.w_bank, self.w_views = fs.working_views(self.dtype)

    def transposed(self, p, w_krsc):
        """Tap-transposed filter for dgrad: from the batched bank when available, else a per-layer transpose."""
        if self.flat_state is not None:
            if self.wt_views is None:
                self.wt_views = self.flat_state.transposed_views(self.w_bank)
            v = self.wt_views.get(id(p))
            if v is not None:
                return v
        return ops.weight_transpose(w_krsc)

    def empty(self, B, H, W, Cn, dtype=None):
        return torch.empty((B, H, W, Cn), dtype=dtype or self.dtype, device=self.device)

    def zeros(self, *shape):
        n = 1
        for d in shape:
            n *= d
        n4 = (n + 3) // 4 * 4                       # keep every carve 16-byte aligned
        self.pool_need += n4
        if self.pool is not None and self.pool_off + n4 <= self.pool.numel():
            t = self.pool[self.pool_off:self.pool_off + n].view(shape)
            self.pool_off += n4
            return t
        return torch.zeros(shape, dtype=torch.float32, device=self.device)


class GradStore:
    """One flat f32 gradient buffer; every parameter's ``.grad`` is a view of it (conv filters in KRSC memory)."""

    def __init__(self, module: torch.nn.Module, order=None):
        self.params = list(order) if order is not None else [p for p in module.parameters() if p.requires_grad]
        self.flat = None
        self.views = {}
        self.offsets = {}             # id(param) -> first element in ``flat``
        self.external_zero = False    # True: the owner (trainer) zeroes ``flat`` itself after each optimizer step

    def _build(self, device):
        pad = lambda n: (n + 7) // 8 * 8                  # same 32-byte parameter alignment as engine/flat.py:padded
        total = sum(pad(p.numel()) for p in self.params)
        self.flat = torch.zeros(total, dtype=torch.float32, device=device)
        off = 0
        for p in self.params:
            n = p.numel()
            seg = self.flat[off:off + n]
            if p.dim() == 4:
                o, i, kh, kw = p.shape
                v = seg.view(o, kh, kw, i).permute(0, 3, 1, 2)     # OIHW shape, channels_last memory
            else:
                v = seg.view(p.shape)
            self.views[id(p)] = v
            self.offsets[id(p)] = off
            off += pad(n)

    def begin_backward(self, device):
        """Attach views; zero the buffer when the optimizer has cleared the grads (fresh accumulation window)."""
        if self.flat is None or self.flat.device != device:
            self._build(device)
        fresh = any(p.grad is None or p.grad.data_ptr() != self.views[id(p)].data_ptr() for p in self.params)
        if fresh:
            if not (self.external_zero and all(p.grad is not None for p in self.params)):
                self.flat.zero_()
            for p in self.params:
                p.grad = self.views[id(p)]

    def grad_krsc(self, p: torch.nn.Parameter) -> torch.Tensor:
        """Contiguous [O][KH][KW][I] f32 view of a conv filter's gradient."""
        return self.views[id(p)].permute(0, 2, 3, 1)

    def grad_vec(self, p: torch.nn.Parameter) -> torch.Tensor:
        return self.views[id(p)]


def engine_dtype(module) -> torch.dtype:
    """Activation dtype: f16 under torch autocast (the reference's AMP, engine/trainer.py:378), else the module's own."""
    if torch.is_autocast_enabled("cuda"):
        return torch.get_autocast_dtype("cuda")
    return getattr(module, "_sy11_dtype", torch.float32)


def to_act(t: torch.Tensor, dtype) -> Act:
    """NCHW-shaped tensor (any strides) -> NHWC Act.  A contiguous f32 3-channel image is also kept raw for the stem."""
    if not t.is_cuda:
        raise ops._lib.Sy11Error("sy11 modules run on the MI355X only: move the model and inputs to 'cuda' "
                                 "(there is no CPU fallback on the hot path)")
    if t.dtype == torch.float32 and t.shape[1] == 3 and t.is_contiguous():
        return Act(None, raw=t, req=t.requires_grad, dtype=dtype)
    v = t.permute(0, 2, 3, 1)
    if v.dtype != dtype or not v.is_contiguous():
        n = torch.empty(v.shape, dtype=dtype, device=t.device)
        n.copy_(v)
        v = n
    return Act(v, req=t.requires_grad)


def from_act(a: Act) -> torch.Tensor:
    """NHWC Act -> NCHW-shaped tensor (channels_last strides, zero copy); non-4-d outputs pass through."""
    return a.data.permute(0, 3, 1, 2) if a.data.dim() == 4 else a.data


def _flatten(x):
    if isinstance(x, (list, tuple)):
        return list(x), True
    return [x], False


class EngineFn(torch.autograd.Function):
    """Runs ``module._run`` on Acts; backward replays the tape and fills parameter grads in the GradStore."""

    @staticmethod
    def forward(ctx, module, n_in, is_list, record, dtype, *tensors):
        ins = tensors[:n_in]
        store = None
        if record:
            store = module.__dict__.get("_sy11_grads")
            if store is None:
                store = GradStore(module)
                module.__dict__["_sy11_grads"] = store
        ec = Ctx(module.training, record, dtype, ins[0].device, store, module.__dict__.get("_sy11_pool_hint", 0))
        ec.attach_flat(module)
        acts = [to_act(t, dtype) for t in ins]
        out = module._run(ec, acts if is_list else acts[0])
        if ec.bn_counters:
            torch._foreach_add_(ec.bn_counters, 1)          # num_batches_tracked of every BN in one multi-tensor launch
        ctx.module = module
        outs, out_list = _flatten(out)
        ctx.ec, ctx.acts, ctx.outs, ctx.n_in = ec, acts, outs, n_in
        ctx.n_t = len(tensors)
        ctx.in_req = [t.requires_grad for t in ins]
        return tuple(from_act(a) for a in outs)

    @staticmethod
    def backward(ctx, *gouts):
        ec: Ctx = ctx.ec
        if ec.grads is not None:
            ec.grads.begin_backward(ec.device)
        for a, g in zip(ctx.outs, gouts):
            if g is None or a.data.dim() != 4:
                continue
            gv = g.permute(0, 2, 3, 1)
            if gv.dtype != a.data.dtype or not gv.is_contiguous():
                gv = gv.to(a.data.dtype).contiguous()
            a.set_grad(gv)
        hook = module_post_backward.get(id(ec.grads)) if ec.grads is not None else None
        mark = ec.marks.get("bucket") if (hook is not None and getattr(hook, "staged", False)) else None
        def first_bucket():                         # every closure of the layers >= the split layer has run: their parameter
            ec.join_side()                          # gradients are final -> the data-parallel hook may start reducing them
            hook(ec.grads, 0)
        ec.run_tape(len(ec.tape) - 1, 0, mark, first_bucket if mark is not None else None)
        ec.join_side()
        ec.tape.clear()
        ctx.module.__dict__["_sy11_pool_hint"] = ec.pool_need       # next step: one pooled allocation
        gin = []
        for a, req in zip(ctx.acts, ctx.in_req):
            if req:
                g = a.grad_read().permute(0, 3, 1, 2)
                gin.append(g.float() if g.dtype != torch.float32 else g)
            else:
                gin.append(None)
        if hook is not None:
            hook(ec.grads, 1)
        return (None, None, None, None, None, *gin, *([None] * (ctx.n_t - ctx.n_in)))


# GradStore id -> callable(GradStore, stage): set by the data-parallel wrapper to all-reduce the flat gradient buffer.
# stage 1 = the backward pass is complete; a hook with ``.staged = True`` is also called with stage 0 as soon as the
# gradients of the layers >= the model's bucket layer are final (the reference's DDP buckets, engine/trainer.py:273).
module_post_backward = {}

# Only the capturing thread's calls can invalidate a capture: under the default "global" mode a helper thread of the process
# (RCCL's watchdog polling its events while another rank is still reducing) would abort the capture of the training graphs.
_CAPTURE_MODE = "thread_local"
_MAX_FWD_GRAPHS = 8


class _Graphed:
    """One captured (forward graph, backward graph) pair for a fixed input signature of a module.

    hipGraph replay replaces ~1300 individually launched kernels per training step by two graph launches: the tape's
    Python closures run ONCE, at capture time; afterwards only device work remains.  Everything the closures touch is
    static: inputs are copied into ``static_in``, activations live in the graph's private pool, parameter gradients in
    the GradStore's flat buffer, output gradients are copied into ``static_gout`` before the backward replay.
    """

    def __init__(self, module, xs, is_list, dtype):
        dev = xs[0].device
        store = module.__dict__.get("_sy11_grads")
        if store is None:
            store = GradStore(module)
            module.__dict__["_sy11_grads"] = store
        store.begin_backward(dev)                                   # flat gradient buffer exists at a fixed address
        self.store = store
        self.static_in = [x.detach().clone() for x in xs]
        self.is_list = is_list
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            self.g_fwd = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_fwd, stream=side, capture_error_mode=_CAPTURE_MODE):
                ec = Ctx(True, True, dtype, dev, store, module.__dict__.get("_sy11_pool_hint", 0))
                ec.capturing = True
                ec.attach_flat(module)
                acts = [to_act(t, dtype) for t in self.static_in]
                out = module._run(ec, acts if is_list else acts[0])
                if ec.bn_counters:
                    torch._foreach_add_(ec.bn_counters, 1)
            self.ec, self.acts = ec, acts
            self.outs, _ = _flatten(out)
            self.static_out = [from_act(a) for a in self.outs]
            self.static_gout = [torch.zeros_like(a.data) for a in self.outs]      # NHWC, activation dtype
            hook = module_post_backward.get(id(store))
            mark = ec.marks.get("bucket") if (hook is not None and getattr(hook, "staged", False)) else None
            if mark is not None and not (0 < mark < len(ec.tape)):
                mark = None
            self.g_bwd = torch.cuda.CUDAGraph()
            self.g_bwd2 = None                                      # second half of a bucketed backward (data parallel only)
            with torch.cuda.graph(self.g_bwd, pool=self.g_fwd.pool(), stream=side, capture_error_mode=_CAPTURE_MODE):
                for a, g in zip(self.outs, self.static_gout):
                    if a.data.dim() == 4:
                        a.set_grad(g)
                ec.run_tape(len(ec.tape) - 1, mark if mark is not None else 0)
                ec.join_side()                                      # the wgrad branch joins inside the captured graph
                if mark is None:
                    self.static_gin = [a.grad_read().permute(0, 3, 1, 2) if a.req else None for a in acts]
            if mark is not None:
                # the layers below the split: a graph of their own, so that the first bucket's all-reduce can be issued
                # between the two replays and run beside this one
                self.g_bwd2 = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.g_bwd2, pool=self.g_fwd.pool(), stream=side, capture_error_mode=_CAPTURE_MODE):
                    ec.run_tape(mark - 1, 0)
                    ec.join_side()
                    self.static_gin = [a.grad_read().permute(0, 3, 1, 2) if a.req else None for a in acts]
            ec.tape.clear()
        torch.cuda.current_stream(dev).wait_stream(side)


class _GraphedFwd:
    """A captured forward-only graph (no tape) for inference / validation / `torch.no_grad()` forwards: the predictor's
    ~170 launches per batch become one graph launch.  Outputs live in the graph's private pool and are overwritten by the
    next replay — callers consume them before the next call (postprocess does)."""

    def __init__(self, module, xs, is_list, dtype):
        dev = xs[0].device
        self.static_in = [x.detach().clone() for x in xs]
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            self.g_fwd = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.g_fwd, stream=side, capture_error_mode=_CAPTURE_MODE):
                ec = Ctx(module.training, False, dtype, dev, None, module.__dict__.get("_sy11_pool_hint", 0))
                ec.capturing = True
                ec.attach_flat(module)
                acts = [to_act(t, dtype) for t in self.static_in]
                out = module._run(ec, acts if is_list else acts[0])
                if ec.bn_counters:
                    torch._foreach_add_(ec.bn_counters, 1)
            outs, _ = _flatten(out)
            self.static_out = [from_act(a) for a in outs]
        torch.cuda.current_stream(dev).wait_stream(side)

    def __call__(self, xs):
        for dst, src in zip(self.static_in, xs):
            if dst.data_ptr() != src.data_ptr():
                dst.copy_(src)
        self.g_fwd.replay()
        return tuple(o.detach() for o in self.static_out)


class GraphFn(torch.autograd.Function):
    """Autograd bridge over a captured pair: copy in, replay forward; copy output grads in, replay backward."""

    @staticmethod
    def forward(ctx, entry, n_in, *tensors):
        for dst, src in zip(entry.static_in, tensors[:n_in]):
            if dst.data_ptr() != src.data_ptr():                    # producers may write straight into the static input
                dst.copy_(src)
        entry.g_fwd.replay()
        ctx.entry, ctx.n_in, ctx.n_t = entry, n_in, len(tensors)
        ctx.in_req = [t.requires_grad for t in tensors[:n_in]]
        return tuple(o.detach() for o in entry.static_out)          # fresh tensor objects over the static storage

    @staticmethod
    def backward(ctx, *gouts):
        e = ctx.entry
        e.store.begin_backward(e.static_in[0].device)               # zeroes the flat buffer at the start of a window
        for dst, g in zip(e.static_gout, gouts):
            if g is not None and dst.dim() == 4 and g.data_ptr() != dst.data_ptr():      # the criterion may have written in place
                dst.copy_(g.permute(0, 2, 3, 1))
        hook = module_post_backward.get(id(e.store))
        e.g_bwd.replay()
        if e.g_bwd2 is not None:
            if hook is not None:
                hook(e.store, 0)                                    # layers >= the split are done: reduce them beside the rest
            e.g_bwd2.replay()
        gin = [(g.float() if (req and g is not None) else None) for g, req in zip(e.static_gin, ctx.in_req)]
        if hook is not None:
            hook(e.store, 1)
        return (None, None, *gin, *([None] * (ctx.n_t - ctx.n_in)))


def graph_static_input(module, shape, dtype=torch.float32):
    """The static input tensor of a captured graph entry of ``module`` matching (shape, dtype), or None: a producer that
    writes its output there saves the per-step copy into the graph's input buffer."""
    cfg = module.__dict__.get("_sy11_graph_cfg")
    if not cfg:
        return None
    want_fwd = not (module.training and torch.is_grad_enabled())       # the same rule run_module uses to pick an entry kind
    for key, entry in cfg["entries"].items():
        if (len(key) == 4) != want_fwd:
            continue
        if len(entry.static_in) == 1 and tuple(entry.static_in[0].shape) == tuple(shape) and entry.static_in[0].dtype == dtype:
            return entry.static_in[0]
    return None


def graph_static_gout(module):
    """Static output-gradient tensors (NHWC) of the captured backward graph that belongs to the forward that just ran (a short
    last batch has its own captured pair), or None when that forward was not a graph replay."""
    cfg = module.__dict__.get("_sy11_graph_cfg")
    if not cfg:
        return None
    last = cfg.get("last_train_entry")
    return last.static_gout if last is not None else None


def enable_graphs(module, warmup: int = 2):
    """Opt a module into hipGraph replay of its train-mode forward/backward (static shapes; see _Graphed)."""
    module.__dict__["_sy11_graph_cfg"] = {"warmup": warmup, "seen": {}, "entries": {}, "last_train_entry": None}
    return module


def run_module(module, x):
    """nn.Module.forward for any sy11 module: tensor(s) in, tensor(s) out, differentiable through EngineFn."""
    cfg = module.__dict__.get("_sy11_graph_cfg")
    if cfg is not None and module.training and torch.is_grad_enabled():
        xs, is_list = _flatten(x)
        dtype = engine_dtype(module)
        key = (tuple((tuple(t.shape), t.dtype) for t in xs), dtype)
        entry = cfg["entries"].get(key)
        if entry is None:
            n = cfg["seen"].get(key, 0)
            cfg["seen"][key] = n + 1
            if n >= cfg["warmup"]:
                entry = cfg["entries"][key] = _Graphed(module, xs, is_list, dtype)
        cfg["last_train_entry"] = entry                         # None while this signature is still running eagerly
        if entry is not None:
            params = [p for p in module.parameters()]
            return GraphFn.apply(entry, len(xs), *xs, *params)
    elif cfg is not None and not torch.is_grad_enabled():
        xs, is_list = _flatten(x)
        dtype = engine_dtype(module)
        key = (tuple((tuple(t.shape), t.dtype) for t in xs), dtype, "fwd", module.training)
        entry = cfg["entries"].get(key)
        if entry is None:
            n = cfg["seen"].get(key, 0)
            cfg["seen"][key] = n + 1
            # at most a few forward-only graphs per module: rect validation feeds a different shape per batch, and every
            # captured shape keeps its own activation pool alive
            if n >= cfg["warmup"] and sum(1 for k in cfg["entries"] if len(k) == 4) < _MAX_FWD_GRAPHS:
                entry = cfg["entries"][key] = _GraphedFwd(module, xs, is_list, dtype)
        if entry is not None:
            return entry(xs)
    return _run_module_eager(module, x)


def _run_module_eager(module, x):
    """nn.Module.forward for any sy11 module: tensor(s) in, tensor(s) out, differentiable through EngineFn."""
    xs, is_list = _flatten(x)
    params = [p for p in module.parameters()]
    record = module.training and torch.is_grad_enabled() and any(t.requires_grad for t in (*xs, *params))
    return EngineFn.apply(module, len(xs), is_list, record, engine_dtype(module), *xs, *params)
