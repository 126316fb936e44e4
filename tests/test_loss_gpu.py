"""GPU: fused detection criterion (csrc/loss.hip) vs the oracle (oracle/loss_ref.py = restated utils/loss.py + tal.py):
assignment bit-exact, loss / items within 1e-4, gradients w.r.t. the head maps within 1e-3 of the gradient scale."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import loss_ref

pytestmark = pytest.mark.gpu
DEV = "cuda"


def crit(nc, fused=True):
    from sy11.utils.loss import v8DetectionLoss
    det = SimpleNamespace(stride=torch.tensor([8., 16., 32.]), nc=nc, reg_max=16)
    model = SimpleNamespace(args=SimpleNamespace(box=7.5, cls=0.5, dfl=1.5), model=[det],
                            parameters=lambda: iter([torch.zeros(1, device=DEV)]))
    return v8DetectionLoss(model, fused=fused)


def make_case(B, nc, hw, n_gt, seed):
    g = torch.Generator().manual_seed(seed)
    maps = [torch.randn(B, 64 + nc, h, w, generator=g) * 1.5 for h, w in hw]
    bi, cl, bb = [], [], []
    for b in range(B):
        k = n_gt[b % len(n_gt)]
        for _ in range(k):
            bi.append(float(b))
            cl.append([float(torch.randint(0, nc, (1,), generator=g))])
            cxcy = 0.25 + 0.5 * torch.rand(2, generator=g)
            wh = 0.15 + 0.5 * torch.rand(2, generator=g)
            bb.append(torch.cat((cxcy, wh)).tolist())
    batch = {"batch_idx": torch.tensor(bi), "cls": torch.tensor(cl).view(-1, 1), "bboxes": torch.tensor(bb).view(-1, 4)}
    return maps, batch


@pytest.mark.parametrize("B,nc,hw,n_gt,seed", [
    (2, 80, [(16, 16), (8, 8), (4, 4)], [3, 1], 0),
    (3, 5, [(20, 12), (10, 6), (5, 3)], [4, 0, 2], 1),          # an image without targets, non-square maps
    (2, 2, [(8, 8), (4, 4), (2, 2)], [6], 2),                   # many overlapping boxes -> multi-gt conflicts
    (2, 80, [(8, 8), (4, 4), (2, 2)], [0], 3),                  # a batch without a single target (utils/loss.py:197-198): background BCE only
    (1, 3, [(4, 4), (2, 2), (1, 1)], [2], 4),                   # one image, 21 anchors: fewer candidates per level than topk = 10 at the coarse levels
])
def test_fused_loss_matches_oracle(B, nc, hw, n_gt, seed):
    from sy11 import ops as K
    maps, batch = make_case(B, nc, hw, n_gt, seed)
    # oracle (CPU, autograd)
    om = [m.clone().requires_grad_(True) for m in maps]
    oloss, oitems, (t_labels, t_boxes, t_scores, fg, gt_idx) = loss_ref.detection_loss(om, batch, nc=nc, return_targets=True)
    oloss.backward()
    # device
    c = crit(nc)
    feats = [m.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True) for m in maps]
    loss, items = c(feats, {k: v.to(DEV) for k, v in batch.items()})
    assert abs(loss.item() - oloss.item()) <= 1e-4 * abs(oloss.item()), (loss.item(), oloss.item())
    np.testing.assert_allclose(items.cpu().numpy(), oitems.numpy(), rtol=1e-4, atol=1e-6)
    (loss * 3.0).backward()
    for f, o in zip(feats, om):
        gscale = o.grad.abs().max().item()
        err = (f.grad.cpu() / 3.0 - o.grad).abs().max().item()
        assert err <= 1e-3 * gscale + 1e-7, (err, gscale)
    # assignment bit-exact: re-run the assign stage and compare with the oracle's TAL outputs
    imgsz = torch.tensor(maps[0].shape[2:], dtype=torch.float32) * 8.0
    gt = loss_ref.pack_targets(batch["batch_idx"], batch["cls"], batch["bboxes"], B, imgsz[[1, 0, 1, 0]])
    w = K.det_loss_forward([m.to(DEV).permute(0, 2, 3, 1).contiguous() for m in maps], (8., 16., 32.), nc, gt.to(DEV))
    asg = w.assign.cpu()
    fg = fg.bool()                                  # (the oracle's mask of a target-free batch is a float zeros tensor)
    # An anchor inside a box whose alignment metric is EXACTLY zero (no overlap with its own prediction) ties with the masked-out
    # anchors in select_topk_candidates' torch.topk (tal.py:196-201); whether it is among the ten is the backend's tie order, and it
    # carries weight zero either way (target score 0: no box / DFL term, background BCE).  Such anchors may differ; all others may not.
    dev_fg, weight = asg >= 0, t_scores.sum(-1)
    differs = dev_fg != fg
    assert not (differs & ((weight > 0) | (w.norm.cpu() > 0))).any(), (dev_fg, fg)
    both = dev_fg & fg
    assert torch.equal(asg[both].long(), gt_idx[both])
    assert torch.allclose(w.norm.cpu(), weight, rtol=1e-4, atol=1e-6)


def test_device_labels_with_recycled_addresses_are_repacked_per_batch():
    """Labels that already live on the device: consecutive batches with the SAME number of targets but a different
    per-image distribution (the caching allocator hands the new tensors the old addresses, version 0) must each be packed
    with their own per-image maximum — a stale maximum indexes the packed (B, maxGT, 5) buffer out of bounds."""
    c = crit(80)
    hw = [(16, 16), (8, 8), (4, 4)]
    for n_gt, seed in (([1, 1, 1, 1], 7), ([4, 0, 0, 0], 8), ([0, 2, 2, 0], 9)):       # 4 targets each time: max 1, 4, 2
        maps, batch = make_case(4, 80, hw, n_gt, seed)
        dev_batch = {k: v.to(DEV) for k, v in batch.items()}                             # fresh device tensors every batch
        feats = [m.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_(True) for m in maps]
        loss, items = c(feats, dev_batch)
        oloss, oitems = loss_ref.detection_loss([m.clone() for m in maps], batch, nc=80)
        assert abs(loss.item() - oloss.item()) <= 1e-4 * abs(oloss.item()), (n_gt, loss.item(), oloss.item())
        del dev_batch, feats
    with pytest.raises(Exception):
        crit(80, fused=False)                         # there is no tensor-op criterion in the product


def test_target_packing_kernel_matches_oracle_bit_exact():
    """sy11_det_loss_pack_targets vs the oracle's restated v8DetectionLoss.preprocess (utils/loss.py:194-207): ragged counts,
    images without targets, a single target, many targets per image (> one 64-lane ballot), host and device labels, strided columns."""
    c = crit(6)
    g = torch.Generator().manual_seed(4)
    cases = [(2, [0, 0, 1]), (4, [3, 0, 3, 3, 1]), (3, [2]), (5, [4, 4, 0, 4, 4, 0, 2]),
             (64, torch.randint(0, 64, (700,), generator=g).tolist()), (3, [1] * 150 + [0] * 3)]
    for B, ids in cases:
        n = len(ids)
        bi = torch.tensor(ids, dtype=torch.float32)
        cls = torch.randint(0, 6, (n, 1), generator=g).float()
        box = torch.cat((0.2 + 0.6 * torch.rand(n, 2, generator=g), 0.05 + 0.3 * torch.rand(n, 2, generator=g)), 1)
        scale = torch.tensor([64., 48., 64., 48.])
        ref = loss_ref.pack_targets(bi, cls, box, B, scale)
        t = torch.cat((bi.view(-1, 1), cls, box), 1)
        for targets, idx in ((t, bi), (t.to(DEV), bi.to(DEV))):                  # host labels (dataloader) and device labels (bench)
            got = c.preprocess(targets, B, scale_tensor=(64.0, 48.0), batch_idx=idx)
            assert got.shape == ref.shape and torch.equal(got.cpu(), ref), (B, n)
        got = c.preprocess(t, B, scale_tensor=scale.to(DEV), batch_idx=bi)       # the reference's tensor-valued scale
        assert torch.equal(got.cpu(), ref)
