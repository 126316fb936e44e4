"""Shared body of the f16 parity tests (tests/test_model_gpu.py, tests/test_fusion_gpu.py).

What is compared: the HIP path in its f16 mode (the reference's AMP dtype, engine/trainer.py:378) against the oracle under
``emulate_f16`` — the CPU restatement with the SAME rounding points (16-bit operands and stored activations / gradients,
f32 accumulation), so the two differ by summation order only, not by the quantisation itself.

The criterion is split the way the reference splits it (utils/loss.py:250-258: the task-aligned assignment runs under
``no_grad`` on detached predictions — a DISCRETE function of the logits — and the loss terms are smooth given its output):

  1. the ASSIGNMENT is tested on its own, bit-exact: the device's assign stage on the device's own f32 head logits against
     ``oracle.loss_ref.tal_assign`` on the same logits (copied to the host);
  2. the LOSS and the GRADIENTS are compared with that one assignment PINNED on both sides (``detection_loss(pinned=...)`` in the
     oracle, ``assign`` / ``norm`` of the criterion workspace overwritten on the device), so a last-bit difference in a logit can
     no longer move an anchor across a top-10 boundary on one side only.  One attempt, no retry;
     The smooth part has kinks of its own: the IoU / enclosing-box terms of CIoU take min / max of a predicted and a target edge
     (utils/metrics.py:109-133), so their gradient JUMPS where the two coincide — which is where training drives them.  r03
     located the rare (about 1 step in 12 for a susceptible state) 3-8 % whole-gradient difference between two device steps that
     differ only in the order of f32 atomics at exactly this point (tools/f16_event_locate.py: the first gradient that differs
     is the box branch of the stride-8 head; everything before it, and the loss value, agree to 1e-5).  So the comparison is split
     once more: (2a) the criterion's gradient is compared on IDENTICAL logits (the device's own head maps, f32 on both sides,
     1e-3), and the model's backward pass is compared from the SAME head-map gradient on both sides (the device's), so a kink
     crossed by one side only cannot masquerade as a backward-pass difference;
  3. how many anchors the oracle's OWN logits (16-bit emulation) would have assigned differently is measured and printed: that is
     the whole effect the r02 retry was hiding (r02 log: one pre-trained state in twelve had the whole gradient 3.4 % apart).

The model state is "default initialisation + a few dozen f32 SGD steps of the CPU ORACLE trainer" (``pretrained_state``): at
initialisation every anchor predicts the same box / class logits (Detect.bias_init); a trained state is simply a more realistic
input.  r03 trained the state with the DEVICE's own f32 kernels, so any change of a kernel's summation order handed this test
a different instance (per-tensor worsts of 3.5 %, 5.5 % and 13 % from three builds), and it let a tensor beyond its bar pass when
the device was within twice the emulation's reorder noise or no farther from the f32 gradient than 1.5x the emulation.  r04
(VERDICT r03 #6): the instance is FIXED — produced on the CPU by oracle/train_ref.py from a seed, nothing under test takes part —
and so are the bars: loss within 2e-3; whole gradient within 2e-2 of the f32 oracle gradient and within 3e-2 of the emulation
(cosine >= 0.9995; why both, see the assertion); per-tensor median within 1e-2; EVERY tensor within 5 % of its norm plus an absolute floor (filters in front of a BatchNorm and biases feeding one have an
exactly-zero true gradient: what is measured there is 16-bit rounding noise on both sides) — with ONE named exception, the two filters in
front of SPPF's max-pools at 20 % (see the loop below for the mechanism and the measured figures).  The emulation with the batch
reordered (2b: its own noise) and the plain f32 oracle (2c: the truth both 16-bit computations approximate) are still run, as
PRINTED DIAGNOSTICS for whoever has to explain a failure — they no longer decide anything.
With ordered reductions (tests/conftest.py) the device result is reproducible bit for bit, so no bar carries a run-to-run allowance."""
from types import SimpleNamespace

import numpy as np
import torch

from oracle import loss_ref, yolo11_ref as R

DEV = "cuda"
GAINS = (7.5, 0.5, 1.5)
STRIDES = (8.0, 16.0, 32.0)


def pretrained_state(cfg, layers, nc, nb, sz, steps, seed=11):
    """The model state under test: the constructor's initialisation (torch seed ``seed``) after ``steps`` SGD steps of the CPU ORACLE
    trainer (oracle/train_ref.py: f32, autograd backward, the reference's update rule) on seeded random images / boxes.
    r03 trained this state with the DEVICE's f32 kernels, so any change of a kernel's summation order handed the test a different
    instance; the CPU trainer depends on nothing under test.  Cached per session (a few tens of seconds)."""
    import hashlib
    import os
    import tempfile
    from oracle import train_ref as TR
    from sy11.nn.tasks import DetectionModel
    tag = hashlib.sha1(repr((cfg, nc, nb, sz, steps, seed, torch.__version__)).encode()).hexdigest()[:16]
    path = os.path.join(tempfile.gettempdir(), f"sy11_f16_parity_state_{tag}.pt")
    if os.path.exists(path):
        return torch.load(path)
    torch.manual_seed(seed)
    m0 = DetectionModel(cfg, ch=3, nc=nc, verbose=False)
    sd0 = {k: v.detach().clone() for k, v in m0.state_dict().items()}
    g = torch.Generator().manual_seed(seed)
    n_img = nb * 8                                                      # eight distinct mini-batches, taken in order, wrapping
    imgs = torch.rand(n_img, 3, sz, sz, generator=g)
    n = 3 * n_img
    labels = (torch.arange(n_img).repeat_interleave(3).float(), torch.randint(0, nc, (n, 1), generator=g).float(),
              torch.cat((0.25 + 0.5 * torch.rand(n, 2, generator=g), 0.1 + 0.4 * torch.rand(n, 2, generator=g)), 1))
    st, losses = TR.train(sd0, layers, nc, imgs, labels, nb, steps)
    print(f"f16 parity {cfg}: CPU oracle pre-training, {steps} steps: loss {losses[0]:.1f} -> {losses[-1]:.1f}")
    sd = {k: v.detach().clone() for k, v in st.sd.items()}
    torch.save(sd, path)
    return sd


def device_pretrained_state(cfg, nc, nb, sz, steps, seed=11):
    """The r03 form: the constructor's initialisation after ``steps`` f32 SGD steps ON THE DEVICE.  Not an input of the f16 parity tests any
    more (a kernel change would change it); tests/test_config2_gpu.py and the probes under tools/ use it where a trained state of a model
    too large for the CPU trainer is wanted and nothing is compared against a fixed instance."""
    from sy11.engine.trainer import DetectionTrainer
    from sy11.nn.tasks import DetectionModel
    torch.manual_seed(seed)
    m0 = DetectionModel(cfg, ch=3, nc=nc, verbose=False)
    tr = DetectionTrainer(m0, batch_size=nb, device=DEV, overrides={"amp": False, "nbs": nb, "warmup_epochs": 0}, graphs=False)
    g = torch.Generator().manual_seed(seed)
    for _ in range(steps):
        n = 3 * nb
        b = {"img": torch.rand(nb, 3, sz, sz, generator=g).to(DEV), "batch_idx": torch.arange(nb).repeat_interleave(3).float().to(DEV),
             "cls": torch.randint(0, nc, (n, 1), generator=g).float().to(DEV),
             "bboxes": torch.cat((0.25 + 0.5 * torch.rand(n, 2, generator=g), 0.1 + 0.4 * torch.rand(n, 2, generator=g)), 1).to(DEV)}
        tr.train_step(b)
    return {k: v.detach().cpu().clone() for k, v in tr.model.state_dict().items()}


def oracle_assignment(maps_nchw, batch, nc):
    """TaskAlignedAssigner outputs (oracle) for head maps given as NCHW f32 CPU tensors."""
    with torch.no_grad():
        _, _, tg = loss_ref.detection_loss([m.clone() for m in maps_nchw], batch, nc=nc, return_targets=True)
    return tg


def device_targets(tg, B, A):
    _, _, t_scores, fg, gt_idx = tg
    assign = torch.where(fg, gt_idx, torch.full_like(gt_idx, -1)).to(torch.int32).view(B, A)
    return assign, t_scores.sum(-1).float().view(B, A)


def pinned_device_step(m, batch, nc, loss_scale, pin=None):
    """Forward to the head maps, the criterion stage by stage through the C-ABI wrappers (assignment optionally overwritten by
    ``pin`` = (assign, norm)), backward through the engine.
    -> (loss, gradient dict, f32 head maps on the host, device assign, loss_scale * d loss / d maps as NCHW f32 on the host)."""
    from sy11 import ops as K
    maps = m(batch["img"].to(DEV))                                        # train mode: the three raw maps, autograd-connected
    B = maps[0].shape[0]
    nhwc = [f.permute(0, 2, 3, 1).contiguous() for f in maps]
    hw = maps[0].shape[2:]
    scale = torch.tensor([hw[1], hw[0], hw[1], hw[0]], dtype=torch.float32) * STRIDES[0]
    gt = loss_ref.pack_targets(batch["batch_idx"], batch["cls"], batch["bboxes"], B, scale).to(DEV)
    w = K.det_loss_assign([t.detach() for t in nhwc], STRIDES, nc, gt)
    dev_assign = w.assign.clone()
    if pin is not None:
        w.assign.copy_(pin[0].to(DEV))
        w.norm.copy_(pin[1].to(DEV))
        w.sums.zero_()
        w.sums[0, 0] = w.norm.sum()
    K.det_loss_terms(w)
    tot = w.sums.sum(0)
    tss = tot[0].clamp(min=1.0)
    items = tot[1:4] / tss * torch.tensor(GAINS, device=DEV)
    loss = items.sum() * B
    up = (torch.full((1,), float(loss_scale), device=DEV) / tss).contiguous()
    dmaps = K.det_loss_backward(w, up, GAINS)
    torch.autograd.backward(maps, [d.permute(0, 3, 1, 2) for d in dmaps])
    grads = {k: p.grad.float().cpu() / loss_scale for k, p in m.named_parameters() if p.requires_grad and p.grad is not None}
    return loss.item(), grads, [f.detach().float().cpu() for f in maps], dev_assign.cpu(), [d.permute(0, 3, 1, 2).float().cpu() for d in dmaps]


def run_f16_parity(cfg, layers, nc, nb=16, sz=256, steps=120, loss_scale=64.0, seed=11, pre_nb=8, pre_sz=192):
    from sy11.nn.tasks import DetectionModel
    sd = pretrained_state(cfg, layers, nc, pre_nb, pre_sz, steps, seed=seed)       # (the CPU trainer's batch: small, it only has to leave the initialisation)
    g = torch.Generator().manual_seed(3)
    img = torch.rand(nb, 3, sz, sz, generator=g)
    n = 2 * nb
    batch = {"img": img, "batch_idx": torch.arange(nb).repeat_interleave(2).float(), "cls": torch.randint(0, nc, (n, 1), generator=g).float(),
             "bboxes": torch.cat((0.3 + 0.4 * torch.rand(n, 2, generator=g), 0.15 + 0.4 * torch.rand(n, 2, generator=g)), 1)}

    def model():
        m = DetectionModel(cfg, ch=3, nc=nc, verbose=False)
        m.args = SimpleNamespace(box=GAINS[0], cls=GAINS[1], dfl=GAINS[2])
        m.load_state_dict(sd)
        m._sy11_dtype = torch.float16
        return m.to(DEV).train()

    # (1) the assignment on its own: device stage vs oracle on the SAME (device) logits — bit-exact
    _, _, dev_maps, dev_assign, _ = pinned_device_step(model(), batch, nc, loss_scale)
    tg = oracle_assignment(dev_maps, batch, nc)
    B, A = dev_assign.shape
    pin = device_targets(tg, B, A)
    assert int((pin[0] >= 0).sum()) > 0
    assert torch.equal(dev_assign, pin[0]), f"{int((dev_assign != pin[0]).sum())} anchors assigned differently on identical logits"

    # (2) loss and gradients with that assignment pinned on both sides
    l1, g1, maps1, _, dm1 = pinned_device_step(model(), batch, nc, loss_scale, pin)
    l2, g2, _, _, _ = pinned_device_step(model(), batch, nc, loss_scale, pin)

    # (2a) the criterion's gradient on IDENTICAL logits: the oracle's loss (f32 autograd) on the device's own head maps against the
    # device's d loss / d maps.  Same inputs, so both sit on the same side of every kink of CIoU (see the module docstring).
    leaves = [t.clone().requires_grad_(True) for t in maps1]
    la, _ = loss_ref.detection_loss(leaves, batch, nc=nc, pinned=tg)
    (la * loss_scale).backward()
    crit = max((d - t.grad).norm().item() / t.grad.norm().item() for d, t in zip(dm1, leaves))
    print(f"f16 parity {cfg}: criterion on the device's own logits: loss rel {abs(l1 - la.item()) / abs(la.item()):.2e}, d loss / d maps worst level {crit:.2e}")
    assert abs(l1 - la.item()) <= 1e-4 * abs(la.item()) and crit <= 1e-3, (l1, la.item(), crit)

    # (2') the model's backward: every oracle run below starts from the SAME head-map gradient the device's backward started from
    def oracle_grads(emulate, order=None):
        o = {k: v.detach().clone() for k, v in sd.items()}
        for k, v in o.items():
            if v.dtype.is_floating_point and "running" not in k and ".dfl." not in k:
                v.requires_grad_(True)
        x = img if order is None else img[order]
        if emulate:
            with R.emulate_f16():
                maps = R.forward(o, layers, x, train=True)
        else:
            maps = R.forward(o, layers, x, train=True)
        bt = batch if order is None else dict(batch, batch_idx=torch.argsort(order)[batch["batch_idx"].long()].float())
        loss, _ = loss_ref.detection_loss([t.detach() for t in maps], bt, nc=nc, pinned=tg if order is None else tuple(t[order] for t in tg))
        torch.autograd.backward(maps, [d if order is None else d[order] for d in dm1])
        return loss, {k: v.grad / loss_scale for k, v in o.items() if v.requires_grad and v.grad is not None}, maps

    oloss, og, omaps = oracle_grads(True)
    # (2b) the instance's own 16-bit noise floor: the SAME emulated computation with the batch in another order (identical
    # mathematics; only the order of the BatchNorm batch sums — hence the last f32 bit in front of every 16-bit store — changes)
    perm = torch.randperm(nb, generator=torch.Generator().manual_seed(5))
    ploss, ng, _ = oracle_grads(True, perm)
    # (2c) the plain f32 oracle: each side's own 16-bit error is its distance from this
    _, tg32, _ = oracle_grads(False)

    # (3) what pinning removed: anchors the oracle's own (emulated 16-bit) logits would assign differently
    own = device_targets(oracle_assignment([t.detach() for t in omaps], batch, nc), B, A)
    flips = int((own[0] != pin[0]).sum())
    n_fg = int((pin[0] >= 0).sum())

    loss_rel = abs(l1 - oloss.item()) / abs(oloss.item())
    keys = [k for k in g1 if k in og]
    assert len(keys) >= 0.95 * len(og)
    fa, fb = torch.cat([g1[k].flatten() for k in keys]), torch.cat([og[k].flatten() for k in keys])
    whole = (fa - fb).norm().item() / fb.norm().item()
    cos = torch.dot(fa.double(), fb.double()).item() / (fa.double().norm().item() * fb.double().norm().item())
    fc = torch.cat([g2[k].flatten() for k in keys])
    rerun = (fa - fc).norm().item() / fc.norm().item()
    fn = torch.cat([ng[k].flatten() for k in keys])
    noise = (fn - fb).norm().item() / fb.norm().item()
    print(f"f16 parity {cfg}: the oracle's own reorder noise: loss rel {abs(ploss.item() - oloss.item()) / abs(oloss.item()):.2e}, whole gradient {noise:.3e}")
    print(f"f16 parity {cfg}: loss rel {loss_rel:.2e}, whole gradient {whole:.3e} (identical device rerun {rerun:.3e}), cosine {cos:.6f}; "
          f"assignment pinned: {n_fg} foreground anchors, {flips} would flip under the oracle's own 16-bit logits")
    from sy11 import _lib
    if _lib.get_option("deterministic"):                    # ordered reductions (tests/conftest.py): an identical rerun is bit-identical
        assert l1 == l2 and rerun == 0.0, (l1, l2, rerun)
    top = sorted(keys, key=lambda k: -(g1[k] - og[k]).norm().item())[:6]     # where a whole-gradient deviation sits, for the log
    print(f"f16 parity {cfg}: largest per-tensor distances (device-oracle, oracle reorder noise, device rerun, tensor norm): "
          + "; ".join(f"{k} {(g1[k] - og[k]).norm().item():.3e} {(ng[k] - og[k]).norm().item():.3e} {(g1[k] - g2[k]).norm().item():.3e} "
                      f"{og[k].norm().item():.3e}" for k in top))
    ft = torch.cat([tg32[k].flatten() for k in keys])
    e_dev, e_orc = (fa - ft).norm().item() / ft.norm().item(), (fb - ft).norm().item() / ft.norm().item()
    print(f"f16 parity {cfg}: distance from the f32 oracle gradient: device f16 {e_dev:.3e}, emulating oracle {e_orc:.3e}")
    # FIXED bars on a FIXED instance (r04): the distances to the reordered emulation (2b) and to the f32 oracle (2c) are
    # printed diagnostics, no longer alternative ways to pass
    # Whole gradient: within 2e-2 of the f32 ORACLE gradient — the truth — and within 3e-2 of the emulation.  Why not "1e-2 of the emulation":
    # on the fusion variant the emulation ITSELF sits 2.6e-2 from the f32 gradient and the device 1.0e-2 (r04, this instance): the
    # reference's autocast, which the emulation follows, rounds after every elementwise operation of GCT / WeightedSpatialAttention,
    # the device's fused kernels round once.  A bar against the emulation alone would fail the device for being closer to the truth.
    # (yolo11n, this instance: device 1.3e-2 / emulation 1.2e-2 from f32, 6.8e-3 apart.)
    assert loss_rel <= 2e-3, (l1, oloss.item())
    assert e_dev <= 2e-2 and whole <= 3e-2 and cos >= 0.9995, (whole, cos, e_dev, e_orc)
    gmax = max(og[k].norm().item() for k in keys)
    rms = fb.norm().item() / len(keys) ** 0.5             # root-mean-square tensor norm: the scale of "a typical tensor"
    # Per tensor: 5 % of its own norm (r03, deterministic: the filters of the 20x20 stages of yolo11n sit at 3.0-3.5 %, everything
    # else below 2 %) plus an absolute floor of 0.2 % of a typical tensor's norm.  Device and emulation round at
    # the same STORES, but not after the same partial sums: a gradient that reaches a concat slice from two consumers is rounded
    # after each accumulation on the device and once, after the f32 sum, in autograd — per-element errors of ~1e-4 that a
    # reduction with cancellation (BatchNorm scale gradients: sum of dz * xhat over all pixels) turns into a few percent of a
    # SMALL result.  The floor bounds that in absolute terms; whole-gradient and median bars above / below stay relative.
    floor = max(1e-4 * gmax, 2e-3 * rms)
    # ONE stated exception to the 5 % bar: the two filters in front of SPPF's max-pools (model.8.cv2 and model.9.cv1 in both YAMLs).  A 5x5
    # max-pool routes a gradient to whichever of two near-equal 16-bit activations wins; two 16-bit computations that round a
    # pre-activation differently pick different winners, and the filters just upstream see it undiluted: the EMULATION run twice with the
    # batch reordered sits 6 % (model.8.cv2) and 9-13 % (model.9.cv1) from itself there on these instances (printed below), the device
    # 1.4 % from the f32 gradient on yolo11n.  Their bar is 20 %; every other tensor keeps 5 %.
    loose = ("model.8.cv2.conv.weight", "model.9.cv1.conv.weight")
    rel, bad = [], []
    for k in keys:
        d = (g1[k] - og[k]).norm().item()
        rel.append(d / (og[k].norm().item() + 1e-4 * gmax))
        own = (ng[k] - og[k]).norm().item()              # the oracle against itself, batch reordered (2b)
        ed, eo = (g1[k] - tg32[k]).norm().item(), (og[k] - tg32[k]).norm().item()      # 16-bit error of either side (2c)
        if d > (0.20 if k in loose else 0.05) * og[k].norm().item() + floor or (k in loose and d > 0.05 * og[k].norm().item() + floor):
            nk = og[k].norm().item()
            print(f"f16 parity {cfg}: {k} is {d / nk:.3e} of its norm from the emulation; the emulation's own reorder noise there is "
                  f"{own / nk:.3e}; distance from the f32 gradient: device {ed / nk:.3e}, emulation {eo / nk:.3e}")
            if d > (0.20 if k in loose else 0.05) * nk + floor:
                bad.append((k, d, nk, own, ed, eo))
    print(f"f16 parity {cfg}: per-tensor worst {max(rel):.3e}, median {float(np.median(rel)):.3e}; largest tensor norm {gmax:.3e}, rms tensor norm {rms:.3e}")
    assert not bad, (floor, bad[:8])
    assert float(np.median(rel)) <= 1e-2, float(np.median(rel))
    return {"loss_rel": loss_rel, "whole": whole, "cos": cos, "median": float(np.median(rel)), "max": max(rel), "flips": flips, "rerun": rerun}
