"""CPU: host-side logic of the product (graph builder, strides, state_dict layout, optimizer groups, producer spec,
engine gradient bookkeeping) and the N>1 path over gloo with world_size 2."""
import os
import subprocess
import sys
import textwrap
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import stft_ref as S, yolo11_ref as R

ROOT = Path(__file__).resolve().parents[1]


def test_parse_model_matches_reference_inventory():
    from sy11.nn.tasks import DetectionModel
    for scale, n_params in (("n", 2624080), ("s", 9458752)):
        m = DetectionModel(f"yolo11{scale}.yaml", verbose=False)
        assert sum(p.numel() for p in m.parameters()) == n_params           # cfg/models/11/yolo11.yaml:10-11
        assert m.stride.tolist() == [8.0, 16.0, 32.0]
        sd = m.state_dict()
        assert set(sd) == set(R.empty_state_dict(R.resolve_graph(scale, nc=80)))
        assert len(sd) == 499
        assert m.save == [4, 6, 10, 13, 16, 19, 22]
        bn = m.model[0].bn
        assert bn.eps == 1e-3 and bn.momentum == 0.03                        # torch_utils.py:417-418
        # constructor state of the reference (its stride dry run on a zero image, tasks.py:359-367): one momentum-0.1 update
        bns = [b for b in m.modules() if type(b) is torch.nn.BatchNorm2d]
        assert len(bns) == 81 and all(int(b.num_batches_tracked) == 1 and float(b.running_mean.abs().max()) == 0.0
                                      and torch.equal(b.running_var, torch.full_like(b.running_var, 0.9)) for b in bns)
    det = m.model[-1]
    assert det.cv2[0][-1].bias.data.eq(1.0).all()                            # head.py:138
    assert abs(det.cv3[0][-1].bias.data[0].item() - np.log(5 / 80 / (640 / 8) ** 2)) < 1e-6


def test_filters_live_in_channels_last_memory():
    from sy11.nn.modules import Conv
    c = Conv(16, 32, 3)
    assert c.conv.weight.permute(0, 2, 3, 1).is_contiguous()
    sd = {k: v.clone() for k, v in c.state_dict().items()}
    c.load_state_dict(sd)
    assert c.conv.weight.permute(0, 2, 3, 1).is_contiguous() and c.conv.weight.shape == (32, 16, 3, 3)


def test_fuse_conv_and_bn_matches_oracle():
    from sy11.nn.modules import Conv
    from sy11.utils.torch_utils import fuse_conv_and_bn, initialize_weights
    c = Conv(8, 12, 3)
    initialize_weights(c)
    sd = {k: R.closed_form("f." + k, tuple(v.shape)) if v.dtype.is_floating_point else v for k, v in c.state_dict().items()}
    c.load_state_dict(sd)
    fused = fuse_conv_and_bn(c.conv, c.bn)
    ref = R.fuse_state_dict({"x." + k: v for k, v in sd.items()})
    assert torch.allclose(fused.weight, ref["x.conv.weight"], atol=1e-6)
    assert torch.allclose(fused.bias, ref["x.conv.bias"], atol=1e-6)


def test_optimizer_groups_follow_reference_rules():
    from sy11.engine.trainer import DetectionTrainer
    from sy11.nn.tasks import DetectionModel
    m = DetectionModel("yolo11n.yaml", verbose=False)
    for k, v in m.named_parameters():
        if ".dfl" in k:
            v.requires_grad = False
    opt = DetectionTrainer.build_optimizer(m, "SGD", 0.01, 0.937, 5e-4)
    bias, decay, norm = (g["params"] for g in opt.param_groups)
    names = {id(p): k for k, p in m.named_parameters()}
    assert all("bias" in names[id(p)] for p in bias)
    assert all(names[id(p)].endswith("bn.weight") for p in norm)
    assert all(names[id(p)].endswith("weight") and "bn" not in names[id(p)] for p in decay)
    assert opt.param_groups[1]["weight_decay"] == 5e-4 and opt.param_groups[0].get("weight_decay", 0) == 0
    assert sum(len(g["params"]) for g in opt.param_groups) == sum(p.requires_grad for p in m.parameters())


def test_producer_filter_bank_is_the_oracles_spec():
    from sy11.data.spectrogram import SpectrogramProducer
    s, w = SpectrogramProducer.filter_bank(S.N_MEL, S.N_FFT, S.WARP_ALPHA, S.MEL_TAPS)
    s2, w2 = S.mel_table()
    assert np.array_equal(s, s2) and np.array_equal(w, w2)
    M = S.mel_matrix()
    assert ((M > 0).sum(0) <= 2).all()                 # every FFT bin feeds at most two filters
    assert (M.sum(1) > 0).all()                        # no empty filter


def test_stft_oracle_linearity_and_parseval_like_property():
    iq = S.synthetic_iq(1, seed=2)[:, :S.N_FFT + 3 * S.HOP]
    win = torch.hann_window(S.N_FFT, periodic=True)
    X = torch.stft(iq, S.N_FFT, hop_length=S.HOP, window=win, center=False, onesided=False, return_complex=True)
    X2 = torch.stft(2 * iq, S.N_FFT, hop_length=S.HOP, window=win, center=False, onesided=False, return_complex=True)
    assert torch.allclose(X2, 2 * X, rtol=1e-5, atol=1e-5)
    fr = iq[0, :S.N_FFT] * win
    assert abs((X[0, :, 0].abs() ** 2).sum().item() - S.N_FFT * (fr.abs() ** 2).sum().item()) < 1e-2 * (X[0, :, 0].abs() ** 2).sum().item()


def test_gradstore_views_and_accumulation_window():
    from sy11.engine import GradStore
    from sy11.nn.modules import Conv
    m = Conv(8, 16, 3)
    gs = GradStore(m)
    gs.begin_backward(torch.device("cpu"))
    assert gs.flat.numel() == sum(p.numel() for p in m.parameters())
    w = m.conv.weight
    assert w.grad.shape == w.shape and gs.grad_krsc(w).is_contiguous() and gs.grad_krsc(w).shape == (16, 3, 3, 8)
    gs.flat.fill_(1.0)
    gs.begin_backward(torch.device("cpu"))             # grads still attached -> accumulation continues, no zeroing
    assert gs.flat.sum().item() == gs.flat.numel()
    m.zero_grad(set_to_none=True)
    gs.begin_backward(torch.device("cpu"))             # optimizer cleared the grads -> fresh window
    assert gs.flat.abs().sum().item() == 0 and w.grad is not None


def test_act_gradient_bookkeeping():
    from sy11.engine import Act
    root = Act(torch.zeros(1, 2, 2, 8))
    a, b = root.slice(0, 4), root.slice(4, 8)
    g, acc = a.grad_for_write()                        # slice first: buffer zeroed, accumulate
    assert acc and g.shape[-1] == 4 and root._grad.abs().sum() == 0
    g2, acc2 = root.grad_for_write()
    assert acc2
    other = Act(torch.zeros(1, 2, 2, 8))
    g3, acc3 = other.grad_for_write()                  # whole buffer first: overwrite
    assert not acc3
    assert other.slice(2, 6).grad_for_write()[1]


DDP_SCRIPT = textwrap.dedent("""
    import os, sys, torch, torch.distributed as dist
    sys.path.insert(0, os.path.join(r"{root}", "spectrogram-yolov11_amd"))
    from sy11.engine import GradStore, ddp, module_post_backward
    rank, local, world = ddp.setup_process_group("gloo")
    assert world == 2
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Conv2d(4, 8, 3), torch.nn.BatchNorm2d(8))
    if rank == 1:
        for p in m.parameters():
            p.data.add_(1.0)
    ddp.broadcast_parameters(m)
    ref = [p.detach().clone() for p in m.parameters()]
    ddp.attach(m)
    gs = m.__dict__["_sy11_grads"]
    gs.begin_backward(torch.device("cpu"))
    gs.flat.copy_(torch.arange(gs.flat.numel(), dtype=torch.float32) * (rank + 1))
    module_post_backward[id(gs)](gs, 1)                    # what EngineFn.backward calls at its end
    expect = torch.arange(gs.flat.numel(), dtype=torch.float32) * 3   # rank0 (x1) + rank1 (x2): SUM, not mean
    assert torch.equal(gs.flat, expect), (gs.flat[:4], expect[:4])
    assert all(p.grad is not None and p.grad.data_ptr() == gs.views[id(p)].data_ptr() for p in m.parameters())
    got = [torch.zeros_like(r) for r in ref]
    for g_, r in zip(got, ref):
        g_.copy_(r)
        dist.broadcast(g_, 0)
        assert torch.equal(g_, r)                          # parameters identical on both ranks after the broadcast
    # flat layout: two broadcasts of the flat buffers + the tensors outside them (integer BN counters)
    from sy11.engine.flat import FlatState
    torch.manual_seed(0)
    m2 = torch.nn.Sequential(torch.nn.Conv2d(4, 8, 3, bias=False), torch.nn.BatchNorm2d(8), torch.nn.Conv2d(8, 18, 1))
    fs = FlatState(m2)
    m2.__dict__["_sy11_flat"] = fs
    if rank == 1:
        fs.flat.add_(3.0); fs.flat_buf.add_(2.0); m2[1].num_batches_tracked.add_(7)
    ddp.broadcast_parameters(m2)
    mine = torch.cat([fs.flat, fs.flat_buf, m2[1].num_batches_tracked.float().view(1)])
    theirs = mine.clone()
    dist.broadcast(theirs, 0)
    assert torch.equal(mine, theirs) and int(m2[1].num_batches_tracked) == 0
    assert m2[0].weight.data_ptr() >= fs.flat.data_ptr() and not m2[0].weight.is_contiguous()      # a permuted view of the flat buffer
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
""")


def test_ddp_flat_allreduce_gloo_world2(tmp_path):
    script = tmp_path / "ddp2.py"
    script.write_text(DDP_SCRIPT.format(root=str(ROOT)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                        "127.0.0.1", "--master-port", "29577", str(script)], capture_output=True, text=True, env=env, timeout=240)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert r.stdout.count("ok") == 2


def test_criterion_refuses_cpu_tensors_loudly():
    """v8DetectionLoss is HIP only — criterion AND target packing (sy11_det_loss_pack_targets since r03; its parity test against the
    oracle's pack_targets is tests/test_loss_gpu.py): CPU tensors must raise, there is no tensor-op detour."""
    from oracle import loss_ref
    from sy11._lib import Sy11Error
    from sy11.utils import tal
    from sy11.utils.loss import v8DetectionLoss
    nc = 6
    det = SimpleNamespace(stride=torch.tensor([8., 16., 32.]), nc=nc, reg_max=16)
    model = SimpleNamespace(args=SimpleNamespace(box=7.5, cls=0.5, dfl=1.5), model=[det],
                            parameters=lambda: iter([torch.zeros(1)]))
    crit = v8DetectionLoss(model)
    g = torch.Generator().manual_seed(4)
    with pytest.raises(Sy11Error):
        crit.preprocess(torch.rand(3, 6), 2, scale_tensor=(64.0, 48.0))
    assert crit.preprocess(torch.zeros(0, 6), 3, scale_tensor=torch.ones(4)).shape == (3, 0, 5)
    maps = [torch.randn(2, 64 + nc, h, h) for h in (8, 4, 2)]
    batch = {"batch_idx": torch.tensor([0., 0., 1.]), "cls": torch.tensor([[1.], [4.], [2.]]),
             "bboxes": torch.tensor([[0.4, 0.5, 0.5, 0.4], [0.6, 0.6, 0.3, 0.6], [0.5, 0.5, 0.8, 0.5]])}
    with pytest.raises(Sy11Error):
        crit(maps, batch)                                   # CPU maps: loud, no tensor-op detour
    with pytest.raises(Sy11Error):
        v8DetectionLoss(model, fused=False)
    # anchor / codec helpers kept under the reference's names
    pts, st = tal.make_anchors([(8, 6), (4, 3)], [8, 16])
    rp, rs = R.make_anchors([(8, 6), (4, 3)], [8, 16])
    assert torch.equal(pts, rp) and torch.equal(st, rs)
    d = torch.rand(5, 66, 4, generator=g) * 3
    a = torch.rand(66, 2, generator=g) * 10
    for xywh in (True, False):
        assert torch.equal(tal.dist2bbox(d, a, xywh=xywh), R.dist2bbox(d, a, xywh=xywh))
    bb = tal.dist2bbox(d, a, xywh=False)
    assert torch.equal(tal.bbox2dist(a, bb, 15), loss_ref.bbox2dist(a, bb, 15))


def test_nms_wrapper_candidate_selection_matches_reference_rows(monkeypatch):
    """non_max_suppression (product, batched) hands the same rows to its suppression core as the reference handed to
    torchvision.ops.nms — the product passes each image's rows already in (score descending, candidate order) order, i.e. the
    golden rows under a stable descending sort of their scores."""
    from sy11.utils import ops as uops
    from tests._golden import load
    gold = load("nms_inputs.npz")
    pred = torch.from_numpy(gold["pred"])
    seen = []

    def fake(boxes, scores, counts, thr, max_keep):
        lo = 0
        for n in counts:                                              # one record per image with candidates, as the reference calls nms
            if n:
                seen.append((boxes[lo:lo + n].clone(), scores[lo:lo + n].clone()))
            lo += n
        return torch.ones(boxes.shape[0], dtype=torch.bool)
    monkeypatch.setattr(uops, "_suppress", fake)
    for tag, kw in (("best", dict(conf_thres=0.25, iou_thres=0.7, multi_label=False)),
                    ("multi", dict(conf_thres=0.05, iou_thres=0.7, multi_label=True))):
        seen.clear()
        out = uops.non_max_suppression(pred.clone(), max_det=300, **kw)
        assert len(seen) == int(gold[f"{tag}.n"]) and len(out) == pred.shape[0]
        for i, (b, s) in enumerate(seen):
            gs = torch.from_numpy(gold[f"{tag}.{i}.scores"])
            order = torch.sort(gs, descending=True, stable=True).indices
            assert np.array_equal(s.numpy(), gs[order].numpy()) and np.array_equal(b.numpy(), gold[f"{tag}.{i}.boxes"][order.numpy()])


def test_nms_wrapper_apriori_labels_and_class_filter_on_cpu(monkeypatch):
    """ops.py:272-278 (apriori labels appended to an image's candidates) and the `classes` filter, with the suppression core stubbed
    (keep everything): the label rows come out with score 1.0 and their class, ahead of every prediction of that image."""
    from sy11.utils import ops as uops
    monkeypatch.setattr(uops, "_suppress", lambda b, s, c, t, k: torch.ones(b.shape[0], dtype=torch.bool))
    g = torch.Generator().manual_seed(0)
    pred = torch.rand(2, 4 + 3, 50, generator=g)
    pred[:, :2] = 20 + 60 * pred[:, :2]
    pred[:, 2:4] = 5 + 20 * pred[:, 2:4]
    labels = [torch.tensor([[2.0, 50.0, 50.0, 10.0, 20.0]]), torch.zeros(0, 5)]
    out = uops.non_max_suppression(pred.clone(), 0.5, 0.6, labels=labels, multi_label=False)
    assert torch.equal(out[0][0], torch.tensor([45.0, 40.0, 55.0, 60.0, 1.0, 2.0]))        # the label: xyxy box, score 1, class 2
    plain = uops.non_max_suppression(pred.clone(), 0.5, 0.6, multi_label=False)
    assert out[0].shape[0] == plain[0].shape[0] + 1 and torch.equal(out[1], plain[1]) and torch.equal(out[0][1:], plain[0])
    only1 = uops.non_max_suppression(pred.clone(), 0.5, 0.6, classes=[1], multi_label=True)
    for a, b in zip(only1, uops.non_max_suppression(pred.clone(), 0.5, 0.6, multi_label=True)):
        assert torch.equal(a, b[b[:, 5] == 1])


def test_flat_state_alignment_and_views():
    """FlatState / GradStore share one padded layout: every parameter starts on an 8-element boundary (the 18-element
    spatial-attention filter of the fusion variant must not misalign its successors), views alias the flat buffers."""
    import torch.nn as nn
    from sy11.engine import GradStore
    from sy11.engine.flat import FlatState, padded
    from sy11.nn.modules import Conv, Fusion

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.f = Fusion([128, 128], "ESChannel")
            self.c = Conv(8, 16, 3)
    m = Net()
    ref = {k: v.clone() for k, v in m.state_dict().items()}
    fs = FlatState(m)
    assert all(off % 8 == 0 for off in fs.offsets.values())
    assert fs.flat.numel() == sum(padded(p.numel()) for p in fs.order) > sum(p.numel() for p in fs.order)
    for k, v in m.state_dict().items():                       # values preserved, parameters now alias the flat buffer
        assert torch.equal(v, ref[k]), k
    w = m.f.sab.cv1.weight
    a = fs.offsets[id(w)]
    assert w.shape == (1, 2, 3, 3) and torch.equal(fs.flat[a:a + 18].view(1, 3, 3, 2).permute(0, 3, 1, 2), w.data)
    fs.flat[a] += 1.0
    assert abs(w.data.flatten()[0].item() - (ref["f.sab.cv1.weight"].flatten()[0].item() + 1.0)) < 1e-6
    gs = GradStore(m, order=fs.order)
    gs.begin_backward(torch.device("cpu"))
    assert gs.flat.numel() == fs.flat.numel()
    assert gs.grad_vec(m.f.gsc2.alpha).reshape(-1).data_ptr() == gs.views[id(m.f.gsc2.alpha)].data_ptr()   # a view, not a copy
    for (s0, s1), g in zip(fs.group_slices, fs.group_tensors(gs.flat)):
        assert g.numel() == s1 - s0


def test_allreduce_hook_defers_inside_an_accumulation_window():
    """ddp.attach's hook reduces the flat gradient buffer only when the trainer says the window is complete."""
    from sy11.engine import GradStore, module_post_backward
    from sy11.engine import ddp
    from sy11.nn.modules import Conv
    m = Conv(8, 16, 3)
    calls = []
    orig = ddp.allreduce_flat
    ddp.allreduce_flat = lambda flat, group=None: calls.append(flat.data_ptr())
    try:
        ddp.attach(m)
        store = m.__dict__["_sy11_grads"]
        store.begin_backward(torch.device("cpu"))
        hook = module_post_backward[id(store)]
        store.defer_allreduce = True
        hook(store, 1)
        assert calls == []
        store.defer_allreduce = False
        hook(store, 1)
        assert calls == [store.flat.data_ptr()]
    finally:
        ddp.allreduce_flat = orig
        module_post_backward.pop(id(m.__dict__["_sy11_grads"]), None)


def test_trainer_warmup_and_lr_schedule_follow_the_reference_rules():
    """engine/trainer.py:330 (nw), :364-377 (per-iteration warm-up), :209-215 + LambdaLR (per-epoch decay), on the host only."""
    import numpy as np
    from sy11.engine.trainer import DetectionTrainer
    from sy11.nn.tasks import DetectionModel
    m = DetectionModel("yolo11n.yaml", nc=80, verbose=False)
    t = DetectionTrainer(m, batch_size=16, device="cpu", overrides={"amp": False}, graphs=False)
    assert t.accumulate == 4                                       # nbs 64 / batch 16
    t.set_schedule(batches_per_epoch=50, epochs=10, lrf=0.01)
    assert t.nw == 150                                             # max(round(3.0 * 50), 100)
    names = [type(t.optimizer).__name__] + [len(g["params"]) for g in t.optimizer.param_groups]
    assert names[0] == "SGD" and len(t.optimizer.param_groups) == 3
    t.ni = 0
    t._warmup()
    lrs = [g["lr"] for g in t.optimizer.param_groups]
    assert abs(lrs[0] - 0.1) < 1e-12 and lrs[1] == 0.0 and lrs[2] == 0.0          # group 0 = biases
    assert all(abs(g["momentum"] - 0.8) < 1e-12 for g in t.optimizer.param_groups) and t.accumulate == 1
    t.ni = 75
    t._warmup()
    exp_other = float(np.interp(75, [0, 150], [0.0, 0.01 * 1.0]))
    assert abs(t.optimizer.param_groups[1]["lr"] - exp_other) < 1e-12
    assert abs(t.optimizer.param_groups[0]["lr"] - float(np.interp(75, [0, 150], [0.1, 0.01]))) < 1e-12
    assert abs(t.optimizer.param_groups[2]["momentum"] - float(np.interp(75, [0, 150], [0.8, 0.937]))) < 1e-12
    assert t.accumulate == max(1, int(np.interp(75, [0, 150], [1, 4]).round()))
    t.ni = 151
    t.accumulate = 4
    t._warmup()                                                    # past warm-up: nothing changes
    assert t.accumulate == 4
    for _ in range(5):
        t.end_epoch()
    assert abs(t.optimizer.param_groups[1]["lr"] - 0.01 * ((1 - 5 / 10) * 0.99 + 0.01)) < 1e-12
    t2 = DetectionTrainer(DetectionModel("yolo11n.yaml", nc=80, verbose=False), batch_size=16, device="cpu",
                          overrides={"amp": False, "optimizer": "auto", "iterations": 300}, graphs=False)
    assert type(t2.optimizer).__name__ == "AdamW" and abs(t2.args.lr0 - round(0.002 * 5 / 84, 6)) < 1e-15 and t2.args.warmup_bias_lr == 0.0


# ---------------------------------------------------------------------------------------------- data parallel, world size 2 (gloo)
DDP_TRAINER_SCRIPT = textwrap.dedent("""
    import os, sys, math, torch, torch.distributed as dist
    sys.path.insert(0, os.path.join(r"{root}", "spectrogram-yolov11_amd")); sys.path.insert(0, r"{root}")
    from sy11.engine import ddp, module_post_backward
    from sy11.engine.trainer import DetectionTrainer
    from sy11.nn.tasks import DetectionModel
    rank, local, world = ddp.setup_process_group("gloo")
    assert world == 2
    torch.manual_seed(1 + rank)                                   # different initial weights per rank: the broadcast must fix that
    m = DetectionModel("yolo11n.yaml", nc=80, verbose=False)
    tr = DetectionTrainer(m, batch_size=16, device="cpu", overrides={{"amp": False, "nbs": 64, "warmup_epochs": 0}}, world_size=world, graphs=False)
    assert tr.accumulate == 2 and tr.flat is not None           # nbs 64 / (16 x 2 ranks)
    store = tr.grad_store
    hook = module_post_backward[id(store)]
    assert hook.staged and tr.model.__dict__["_sy11_bucket_layer"] == ddp.BUCKET_LAYER
    w0 = tr.flat.flat.clone()
    ref = w0.clone(); dist.broadcast(ref, 0)
    assert torch.equal(w0, ref)                                   # rank-0 weights everywhere (DDP constructor semantics)
    n = store.flat.numel()
    base = torch.linspace(-1.0, 1.0, n)

    class FakeBackward(torch.autograd.Function):                 # what EngineFn.backward does, minus the kernels: fill the flat
        @staticmethod                                            # gradient views, call the hook at the bucket mark and at the end
        def forward(ctx, w, step):
            ctx.step = step
            return w.sum() * 0.0 + 1.0
        @staticmethod
        def backward(ctx, g):
            store.begin_backward(torch.device("cpu"))
            store.flat.add_(base * (1.0 + rank) * (1.0 + 0.5 * ctx.step) * float(g))
            hook(store, 0)
            hook(store, 1)
            return None, None

    steps = []
    tr.model.forward = lambda batch: (FakeBackward.apply(tr.flat_params[0], float(len(steps))), torch.zeros(3))
    tr.preprocess_batch = lambda b: b
    # expected: window of 2 micro-steps, gradient = SUM over ranks (x1 + x2) of the per-rank sums, then clip 10, SGD-nesterov
    for it in range(4):
        before = tr.flat.flat.clone()
        tr.train_step({{}})
        steps.append(it)
        if it % 2 == 0:
            assert torch.equal(tr.flat.flat, before) and store.flat.abs().sum() > 0          # window open: no step, no reduction
            expect_local = base * (1.0 + rank) * (1.0 + 0.5 * it)
            assert torch.allclose(store.flat, expect_local, atol=1e-6)                        # NOT reduced yet (deferred)
        else:
            assert not torch.equal(tr.flat.flat, before) and store.flat.abs().sum() == 0     # stepped, gradients cleared
    # closed form of the two optimizer steps on every rank
    w = w0.clone(); buf = torch.zeros_like(w); mom, lr = tr.args.momentum, tr.args.lr0
    a, b_ = tr.flat.group_slices[0]
    for win in range(2):
        g = base * 3.0 * ((1.0 + 0.5 * (2 * win)) + (1.0 + 0.5 * (2 * win + 1)))             # ranks (1 + 2) x two micro-steps
        # padding elements of the flat layout carry no parameter: the trainer's buffers keep them at zero only if grads there are
        # ignored by nobody -- they are part of the clip norm exactly as in the product (same flat tensors)
        coef = min(10.0 / (float(g.double().norm()) + 1e-6), 1.0)
        g = g * coef
        gd = g.clone(); gd[a:b_] += tr.optimizer.param_groups[1]["weight_decay"] * w[a:b_]
        buf = gd.clone() if win == 0 else buf * mom + gd
        w = w - lr * (gd + mom * buf)
    assert torch.allclose(tr.flat.flat, w, rtol=1e-5, atol=1e-6), (tr.flat.flat - w).abs().max()
    other = tr.flat.flat.clone(); dist.broadcast(other, 0)
    assert torch.equal(other, tr.flat.flat)                       # ranks stay bit-identical
    # bucket ranges partition the flat buffer
    hook._ranges()
    cover = torch.zeros(n)
    for lo, hi in hook.late + hook.early:
        cover[lo:hi] += 1
    assert torch.equal(cover, torch.ones(n)) and len(hook.late) <= 3 and len(hook.early) <= 3
    assert sum(hi - lo for lo, hi in hook.late) > 0.9 * n
    # tuner picks: rank 0's table everywhere
    from sy11 import _lib
    import struct
    if rank == 0:
        _lib.tune_import(struct.pack("<Qii", 4242, 0, 3) + struct.pack("<Qii", 77, 1, 9))
    else:
        _lib.tune_import(struct.pack("<Qii", 1, 0, 1))
    assert ddp.share_tuner_picks() == 2
    assert sorted(_lib.tune_export()[i:i + 16] for i in (0, 16)) == sorted([struct.pack("<Qii", 4242, 0, 3), struct.pack("<Qii", 77, 1, 9)])
    # ... and the trigger is rank-uniform (ADVICE r03): flags that differ across ranks (multi_scale sizes drawn per rank, an uneven
    # last batch) must never leave one rank in the broadcast and the other in the gradient all-reduce
    _lib.load().sy11_tune_clear()
    if rank == 0:
        _lib.tune_import(struct.pack("<Qii", 5, 0, 2))
    for step_flags in ((False, False), (True, False), (False, True), (True, True), (False, False)):
        shared = ddp.share_tuner_picks_if_any(step_flags[rank])
        assert shared == any(step_flags), (step_flags, shared)
        probe = torch.tensor([float(rank + 1)])
        dist.all_reduce(probe)                                   # the next collective of a step lines up on both ranks
        assert probe.item() == 3.0
    assert _lib.tune_export() == struct.pack("<Qii", 5, 0, 2)
    # early stopping: only rank 0 knows the fitness, everybody must leave fit() together (engine/trainer.py:456-461)
    class Loader:
        dataset = None
        def __len__(self): return 1
        def __iter__(self): return iter([{{}}])
    import sy11.engine.validator as V
    seen = []
    def fake_validate(model, batches):
        seen.append(1)
        return {{"fitness": 0.5 - 0.1 * len(seen)}}              # gets worse every epoch
    V.DetectionValidator = lambda *a, **k: fake_validate
    hist = tr.fit(Loader(), epochs=6, val_batches=(lambda: [None]), save_dir=None, patience=2)
    assert len(hist) == 3, len(hist)                              # epoch 0 sets best, epochs 1-2 are stale -> stop on BOTH ranks
    dist.barrier()
    dist.destroy_process_group()
    print("rank", rank, "ok")
""")


def test_ddp_real_trainer_step_accumulate_buckets_and_early_stop_gloo_world2(tmp_path):
    """The N > 1 path on CPU: the REAL DetectionTrainer.train_step / optimizer_step / fit over flat buffers with an accumulation
    window of 2, the two-bucket staged gradient sum over gloo, rank-0 weights and tuner picks everywhere, and the early-stop
    broadcast — two ranks launched by sy11.engine.ddp.launch (the spawn-before-GPU-init launcher of train(device=[...]))."""
    script = tmp_path / "ddp_trainer.py"
    script.write_text(DDP_TRAINER_SCRIPT.format(root=str(ROOT)))
    sys.path.insert(0, str(ROOT / "spectrogram-yolov11_amd"))
    from sy11.engine import ddp
    codes = ddp.launch([str(script)], 2, env=dict(os.environ, OMP_NUM_THREADS="1", SY11_DDP_OVERLAP="1"), timeout=300)
    assert codes == [0, 0]


def test_train_device_list_spawns_one_child_per_gpu(monkeypatch, tmp_path):
    """YOLO(...).train(device=[0, 1]) outside a launcher: one child per GPU runs the same call (utils/dist.py:25-66); checked
    here is the command the launcher receives (no GPUs in this container)."""
    from sy11.engine import ddp
    from sy11.engine.model import YOLO, _device_list
    assert _device_list("0,1") == [0, 1] and _device_list([2, 3]) == [2, 3] and _device_list("cuda:1") == [1] and _device_list(None) == []
    seen = {}

    def fake_launch(args, nproc, env=None, timeout=None):
        seen.update(args=list(args), nproc=nproc, env=env, src=open(args[0]).read())
        (tmp_path / "results.json").write_text('[{"epoch": 0, "metrics": {"fitness": 0.5}}]')     # what rank 0 of the children leaves behind
        return [0] * nproc
    monkeypatch.setattr(ddp, "launch", fake_launch)
    monkeypatch.delenv("RANK", raising=False)
    y = YOLO("yolo11n.yaml", nc=3)
    out = y.train(data={"train": "x", "val": "y", "names": {0: "a", 1: "b", 2: "c"}, "nc": 3}, epochs=2, batch=8, device=[0, 1], save_dir=str(tmp_path), lr0=0.02)
    assert seen["nproc"] == 2 and seen["env"]["HIP_VISIBLE_DEVICES"] == "0,1"
    assert "m.train(**P['kw'])" in seen["src"] and "yolo11n.yaml" in seen["src"] and '\\"lr0\\": 0.02' in seen["src"] and '\\"epochs\\": 2' in seen["src"]
    assert not os.path.exists(seen["args"][0])                    # the temporary launcher file is removed
    assert out == [{"epoch": 0, "metrics": {"fitness": 0.5}}] and y.metrics == {"fitness": 0.5}      # same return type as one process: the history
    from pathlib import Path
    y.train(data={"train": "x", "names": {0: "a"}, "nc": 1}, epochs=1, device=[0, 1], save_dir=Path(tmp_path), resume=Path("last.pt"))   # Path-valued arguments serialise
    import pytest as _pt
    from sy11._lib import Sy11Error
    with _pt.raises(Sy11Error):
        _device_list("mps")
    assert _device_list("cuda") == [0]
