"""CPU: the oracle (oracle/*.py restatement) must reproduce the REFERENCE's numbers in tests/golden/."""
import numpy as np
import pytest
import torch

from oracle import loss_ref, nms_ref, yolo11_ref as R
from tests._golden import check, load

CASES = {
    # name: (fn(sd, prefix, x, train), input shape)
    "conv_k1": (lambda sd, p, x, tr: R.conv_bn_act(sd, p, x, 1, 1, train=tr), (2, 32, 8, 8)),
    "conv_k3": (lambda sd, p, x, tr: R.conv_bn_act(sd, p, x, 3, 1, train=tr), (2, 32, 8, 8)),
    "conv_k3s2": (lambda sd, p, x, tr: R.conv_bn_act(sd, p, x, 3, 2, train=tr), (2, 32, 10, 10)),
    "conv_k3s2_odd": (lambda sd, p, x, tr: R.conv_bn_act(sd, p, x, 3, 2, train=tr), (1, 16, 9, 7)),
    "conv_noact": (lambda sd, p, x, tr: R.conv_bn_act(sd, p, x, 1, 1, act=False, train=tr), (2, 32, 8, 8)),
    "conv_stem": (lambda sd, p, x, tr: R.conv_bn_act(sd, p, x, 3, 2, train=tr), (2, 3, 16, 16)),
    "dwconv": (lambda sd, p, x, tr: R.conv_bn_act(sd, p, x, 3, 1, g=64, train=tr), (2, 64, 8, 8)),
    "bottleneck": (lambda sd, p, x, tr: R.bottleneck(sd, p, x, 64, 64, True, (3, 3), 0.5, tr), (2, 64, 8, 8)),
    "c3k": (lambda sd, p, x, tr: R.c3k(sd, p, x, 64, 2, True, tr), (2, 64, 8, 8)),
    "c3k2_plain": (lambda sd, p, x, tr: R.c3k2(sd, p, x, 128, 1, False, 0.25, True, tr), (2, 64, 8, 8)),
    "c3k2_c3k": (lambda sd, p, x, tr: R.c3k2(sd, p, x, 64, 1, True, 0.5, True, tr), (2, 64, 8, 8)),
    "sppf": (lambda sd, p, x, tr: R.sppf(sd, p, x, 5, tr), (2, 64, 8, 8)),
    "attention": (lambda sd, p, x, tr: R.attention(sd, p, x, 2, 0.5, tr), (2, 128, 6, 5)),
    "psablock": (lambda sd, p, x, tr: R.psablock(sd, p, x, 2, tr), (2, 128, 6, 5)),
    "c2psa": (lambda sd, p, x, tr: R.c2psa(sd, p, x, 1, 0.5, tr), (2, 128, 6, 5)),
}


def module_state(gold, name):
    """Rebuild the closed-form state of a module fixture from the golden's key list."""
    keys = {k[len(name) + len(".train.grad."):-len(".shape")] for k in gold
            if k.startswith(name + ".train.grad.") and k.endswith(".shape")}
    bufs = {k[len(name) + len(".train.buf."):-len(".shape")] for k in gold
            if k.startswith(name + ".train.buf.") and k.endswith(".shape")}
    sd = {}
    for k in keys:
        shape = tuple(gold[f"{name}.train.grad.{k}.shape"].tolist())
        sd[k] = R.closed_form(name + "." + k, shape).requires_grad_(True)
    for k in bufs:
        shape = tuple(gold[f"{name}.train.buf.{k}.shape"].tolist())
        sd[k] = R.closed_form(name + "." + k, shape)
    return sd, keys, bufs


@pytest.mark.parametrize("name", list(CASES))
def test_module_matches_reference(name):
    gold = load("modules.npz")
    fn, shape = CASES[name]
    sd, keys, bufs = module_state(gold, name)
    x = R.closed_form("in." + name, shape, "signed").requires_grad_(True)
    y = fn(sd, "", x, True)
    g = R.closed_form("g." + name, tuple(y.shape), "signed")
    (y * g).sum().backward()
    check(gold, f"{name}.train.y", y)
    check(gold, f"{name}.train.dx", x.grad, rtol=2e-4)
    for k in keys:
        check(gold, f"{name}.train.grad.{k}", sd[k].grad, rtol=5e-4, atol=2e-5)
    for k in bufs:
        check(gold, f"{name}.train.buf.{k}", sd[k])
    with torch.no_grad():   # the reference ran eval AFTER its train step: running stats carry the 0.03 update
        check(gold, f"{name}.eval.y", fn(sd, "", x.detach(), False))


def test_detect_head_matches_reference():
    gold = load("modules.npz")
    keys = {k[len("detect.train.grad."):-len(".shape")] for k in gold
            if k.startswith("detect.train.grad.") and k.endswith(".shape")}
    sd = {}
    for k in keys:
        sd[k] = R.closed_form("detect." + k, tuple(gold[f"detect.train.grad.{k}.shape"].tolist())).requires_grad_(True)
    # BN buffers are not parameters: rebuild from the conv weight shapes
    for k in list(sd):
        if k.endswith("bn.weight"):
            p = k[:-len("weight")]
            c = sd[k].shape[0]
            sd[p + "running_mean"] = R.closed_form("detect." + p + "running_mean", (c,))
            sd[p + "running_var"] = R.closed_form("detect." + p + "running_var", (c,))
    feats = [R.closed_form(f"in.detect.{i}", s, "signed").requires_grad_(True)
             for i, s in enumerate([(2, 32, 8, 8), (2, 64, 4, 4), (2, 128, 2, 2)])]
    maps = R.detect_head(sd, "", feats, 5, train=True)
    tot = 0
    for i, mp in enumerate(maps):
        check(gold, f"detect.train.map{i}", mp)
        tot = tot + (mp * R.closed_form(f"g.detect.{i}", tuple(mp.shape), "signed")).sum()
    tot.backward()
    for i, f in enumerate(feats):
        check(gold, f"detect.train.dx{i}", f.grad, rtol=2e-4)
    for k in keys:
        check(gold, f"detect.train.grad.{k}", sd[k].grad, rtol=5e-4, atol=2e-5)
    with torch.no_grad():   # eval after the train step, as the generator did
        maps = R.detect_head(sd, "", [f.detach() for f in feats], 5, train=False)
        y = R.detect_decode(maps, (8.0, 16.0, 32.0), 5)
    check(gold, "detect.eval.y", y)


def tiny_model():
    layers = R.resolve_graph("t", nc=4)
    sd = R.seeded_state_dict(R.empty_state_dict(layers), seed=0)
    return layers, sd


def tiny_batch(gold):
    return {"img": R.seeded_image((2, 3, 64, 64), seed=5),
            "batch_idx": torch.from_numpy(gold["batch.batch_idx"]),
            "cls": torch.from_numpy(gold["batch.cls"]),
            "bboxes": torch.from_numpy(gold["batch.bboxes"])}


def test_tiny_model_param_inventory():
    layers, sd = tiny_model()
    n = sum(v.numel() for k, v in sd.items() if "running" not in k and "num_batches" not in k)
    # scale t @nc=2 has 810 550 params; @nc=4 adds 3 levels x 2 classes x (c3+1)
    c3 = max(layers[-1]["ch"][0], min(4, 100))
    assert n == 810550 + 3 * 2 * (c3 + 1)
    assert len(sd) == 499
    s = R.resolve_graph("s", nc=80)
    n_s = sum(v.numel() for k, v in R.empty_state_dict(s).items() if "running" not in k and "num_batches" not in k)
    assert n_s == 9458752


def test_tiny_model_train_forward_loss_grads_match_reference():
    gold = load("model_t.npz")
    layers, sd = tiny_model()
    for v in sd.values():
        if v.dtype.is_floating_point:
            v.requires_grad_(True)
    batch = tiny_batch(gold)
    # BN running buffers must not require grad
    for k in sd:
        if "running" in k:
            sd[k] = sd[k].detach()
    maps = R.forward(sd, layers, batch["img"], train=True)
    for i, mp in enumerate(maps):
        check(gold, f"train.map{i}", mp, rtol=2e-4, atol=2e-5)
    loss, items, tgt = loss_ref.detection_loss(maps, batch, nc=4, return_targets=True)
    assert abs(loss.item() - gold["loss"][0]) <= 1e-4 * abs(gold["loss"][0])
    np.testing.assert_allclose(items.double().numpy(), gold["loss_items"], rtol=1e-4)
    t_labels, t_boxes, t_scores, fg, gt_idx = tgt
    assert np.array_equal(fg.numpy(), gold["tal.fg"])            # bit-exact assignment
    assert np.array_equal(gt_idx.numpy(), gold["tal.gt_idx"])
    assert np.array_equal(t_labels.numpy(), gold["tal.labels"])
    check(gold, "tal.scores", t_scores, rtol=1e-4)
    # the reference divides target_bboxes by the stride IN PLACE after the assigner (utils/loss.py:266)
    _, stride_t = R.make_anchors([m.shape[2:] for m in maps], R.STRIDES)
    check(gold, "tal.bboxes", t_boxes / stride_t)
    loss.backward()
    names = [str(n) for n in gold["grad.names"]]
    for n, (gn, gs) in zip(names, gold["grad.norm_sum"]):
        g = sd[n].grad.double()
        assert abs(g.norm().item() - gn) <= 2e-3 * gn + 1e-6, n
    for k in ("model.0.conv.weight", "model.2.m.0.cv1.conv.weight", "model.10.m.0.attn.qkv.conv.weight",
              "model.23.cv2.0.2.bias", "model.23.cv3.2.2.weight", "model.8.m.0.m.1.cv2.bn.weight"):
        check(gold, "grad." + k, sd[k].grad, rtol=2e-3, atol=2e-4)
    for k in ("model.0.bn.running_mean", "model.0.bn.running_var", "model.22.cv2.bn.running_var"):
        check(gold, "buf." + k, sd[k])


def test_tiny_model_eval_and_fused_match_reference():
    gold = load("model_t.npz")
    layers, sd = tiny_model()
    img = R.seeded_image((2, 3, 64, 64), seed=5)
    with torch.no_grad():
        y, maps = R.forward(sd, layers, img, train=False)
        check(gold, "eval.y", y, rtol=2e-4, atol=2e-5)
        for i, mp in enumerate(maps):
            check(gold, f"eval.map{i}", mp, rtol=2e-4, atol=2e-5)
        yf, _ = R.forward(R.fuse_state_dict(sd), layers, img, train=False, fused=True)
        check(gold, "eval_fused.y", yf, rtol=5e-4, atol=5e-5)


@pytest.mark.parametrize("tag,kw", [("best", dict(conf_thres=0.25, multi_label=False)),
                                    ("multi", dict(conf_thres=0.05, multi_label=True))])
def test_nms_wrapper_inputs_match_reference(tag, kw):
    """The rows our wrapper hands to the NMS core equal what the reference hands to torchvision.ops.nms."""
    gold = load("nms_inputs.npz")
    pred = torch.from_numpy(gold["pred"])
    n = int(gold[f"{tag}.n"])
    assert n == pred.shape[0]
    for xi in range(n):
        x = nms_ref.pre_nms(pred[xi].transpose(0, 1), 6, **kw)
        boxes = x[:, :4] + x[:, 5:6] * 7680
        assert np.array_equal(boxes.numpy(), gold[f"{tag}.{xi}.boxes"])
        assert np.array_equal(x[:, 4].numpy(), gold[f"{tag}.{xi}.scores"])


def test_nms_core_properties():
    """Third-party core (parity unpinned): idempotence, score order, no kept pair above threshold, ties."""
    rng = np.random.default_rng(0)
    xy = rng.uniform(0, 200, (400, 2)).astype(np.float32)
    wh = rng.uniform(5, 60, (400, 2)).astype(np.float32)
    boxes = np.concatenate([xy, xy + wh], 1)
    scores = rng.uniform(0, 1, 400).astype(np.float32)
    keep = nms_ref.nms_core(boxes, scores, 0.5)
    assert np.all(np.diff(scores[keep]) <= 0)
    again = nms_ref.nms_core(boxes[keep], scores[keep], 0.5)
    assert np.array_equal(again, np.arange(len(keep)))
    # brute-force check of the greedy definition
    def iou(a, b):
        w = max(0, min(a[2], b[2]) - max(a[0], b[0])); h = max(0, min(a[3], b[3]) - max(a[1], b[1]))
        i = np.float32(w) * np.float32(h)
        return i / ((a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - i)
    for a in range(len(keep)):
        for b in range(a + 1, len(keep)):
            assert iou(boxes[keep[a]], boxes[keep[b]]) <= 0.5
    # every dropped box is suppressed by some higher-scored kept box
    kept = set(keep.tolist())
    for j in range(400):
        if j not in kept:
            assert any(scores[i] >= scores[j] and iou(boxes[i], boxes[j]) > 0.5 for i in keep)
    # ties: identical boxes & scores -> lowest index survives
    b2 = np.tile(np.array([[0, 0, 10, 10]], np.float32), (4, 1))
    assert nms_ref.nms_core(b2, np.ones(4, np.float32), 0.5).tolist() == [0]
    assert nms_ref.nms_core(np.zeros((0, 4), np.float32), np.zeros(0, np.float32), 0.5).shape == (0,)
