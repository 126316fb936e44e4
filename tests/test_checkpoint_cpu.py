"""CPU: checkpoint interchange.  tests/golden/ref_ckpt_t.pt.gz was written by the REFERENCE
(oracle/gen_golden_ckpt.py: pickled ultralytics module objects, f16, the keys of trainer.save_model); it must load into
sy11 classes without ultralytics being importable, with every weight intact; sy11's own checkpoints round-trip."""
import gzip
import io
import sys

import torch

from oracle import yolo11_ref as R
from tests._golden import GOLD


def ref_ckpt_bytes():
    return gzip.open(GOLD / "ref_ckpt_t.pt.gz", "rb").read()


def test_reference_checkpoint_loads_into_sy11_classes():
    from sy11.engine.checkpoint import attempt_load_one_weight, load_checkpoint
    from sy11.nn import modules as M
    from sy11.nn.tasks import DetectionModel
    assert "ultralytics" not in sys.modules
    ckpt = load_checkpoint(io.BytesIO(ref_ckpt_bytes()))
    assert "ultralytics" not in sys.modules
    assert {"epoch", "best_fitness", "model", "ema", "updates", "optimizer", "train_args", "train_metrics", "date", "version"} <= set(ckpt)
    assert ckpt["epoch"] == 3 and ckpt["updates"] == 17 and ckpt["model"] is None
    ema = ckpt["ema"]
    assert type(ema) is DetectionModel and type(ema.model[0]) is M.Conv and type(ema.model[-1]) is M.Detect
    assert type(ema.model[10]) is M.C2PSA and type(ema.model[2]) is M.C3k2 and type(ema.model[9]) is M.SPPF
    sd = ema.state_dict()
    layers = R.resolve_graph("t", nc=2)
    expect = R.pattern_state_dict(R.empty_state_dict(layers))
    assert set(sd) == set(expect)
    for k, v in sd.items():
        if v.dtype.is_floating_point and ".dfl." not in k:
            assert v.dtype == torch.float16
            assert torch.equal(v.float(), expect[k].half().float()), k
    model, _ = attempt_load_one_weight(io.BytesIO(ref_ckpt_bytes()))
    assert not model.training and next(model.parameters()).dtype == torch.float32
    assert model.names == {0: "lte", 1: "nr"} and model.args.box == 7.5 and model.yaml["scale"] == "t"
    assert torch.equal(model.stride, torch.tensor([8.0, 16.0, 32.0]))
    w = model.model[0].conv.weight
    assert w.is_contiguous(memory_format=torch.channels_last) and w._sy11_groups == 1
    assert model.model[-1].cv3[0][0][0].conv.weight._sy11_groups == model.model[-1].cv3[0][0][0].conv.groups > 1   # depthwise branch


def test_unknown_reference_class_is_refused():
    import pickle

    import pytest
    from sy11._lib import Sy11Error
    from sy11.engine.checkpoint import _RefUnpickler

    data = b"cultralytics.nn.modules.block\nGhostBottleneck\n."          # protocol-0 pickle: GLOBAL of a class we do not ship
    with pytest.raises(Sy11Error, match="outside the MI355X hot path"):
        _RefUnpickler(io.BytesIO(data)).load()
    assert pickle.loads(pickle.dumps({"a": 1})) == {"a": 1}


def test_sy11_checkpoint_round_trip():
    from sy11.engine.checkpoint import attempt_load_one_weight, load_checkpoint, save_checkpoint
    from sy11.nn.tasks import DetectionModel
    m = DetectionModel("yolo11n.yaml", nc=3, verbose=False)
    m.names = {0: "a", 1: "b", 2: "c"}
    m.__dict__["_sy11_graph_cfg"] = {"junk": 1}                      # engine caches must not be pickled
    opt = torch.optim.SGD(m.parameters(), lr=0.01, momentum=0.9)
    for p in m.parameters():
        if p.requires_grad:
            p.grad = torch.ones_like(p)
    opt.step()
    buf = io.BytesIO()
    n = save_checkpoint(buf, ema_model=m, updates=5, optimizer=opt, epoch=2, best_fitness=0.5, train_args={"box": 7.5, "cls": 0.5, "dfl": 1.5})
    assert n == len(buf.getvalue()) > 1000
    ckpt = load_checkpoint(io.BytesIO(buf.getvalue()))
    assert ckpt["epoch"] == 2 and ckpt["updates"] == 5 and ckpt["model"] is None
    assert "_sy11_graph_cfg" not in ckpt["ema"].__dict__
    st = next(iter(ckpt["optimizer"]["state"].values()))
    assert st["momentum_buffer"].dtype == torch.float16
    model, _ = attempt_load_one_weight(io.BytesIO(buf.getvalue()))
    for (k, a), (_, b) in zip(m.state_dict().items(), model.state_dict().items()):
        if a.dtype.is_floating_point:
            assert torch.equal(a.half().float(), b), k
    for k, v in ckpt["sy11_state_dict"].items():
        assert v.dtype in (torch.float16, torch.int64, torch.float32)
    assert ckpt["sy11_yaml"]["nc"] == 3


def test_reference_pickle_carrying_a_criterion_loads_and_saves_again():
    """A reference model pickled after its first loss call carries `criterion` (v8DetectionLoss with BboxLoss / DFLoss /
    TaskAlignedAssigner children, nn/tasks.py:288-300).  Its state is the reference's, not this build's: the loader drops it
    (BaseModel.loss rebuilds it) and the loaded model pickles again."""
    import types

    from sy11.engine.checkpoint import attempt_load_one_weight, save_checkpoint
    from sy11.nn.tasks import DetectionModel
    fake = {}
    for modname, names in (("ultralytics.utils.loss", ("v8DetectionLoss", "BboxLoss", "DFLoss")), ("ultralytics.utils.tal", ("TaskAlignedAssigner",))):
        mod = types.ModuleType(modname)
        for n in names:
            setattr(mod, n, type(n, (object,) if n == "v8DetectionLoss" else (torch.nn.Module,), {"__module__": modname}))
        fake[modname] = mod
    for pkg in ("ultralytics", "ultralytics.utils"):                 # parent packages, so that pickle's import of the leaf resolves
        fake[pkg] = types.ModuleType(pkg)
        fake[pkg].__path__ = []
    sys.modules.update(fake)
    try:
        L, T = fake["ultralytics.utils.loss"], fake["ultralytics.utils.tal"]
        m = DetectionModel("yolo11n.yaml", nc=3, verbose=False)
        crit = L.v8DetectionLoss()
        bbox = L.BboxLoss()
        bbox.dfl_loss = L.DFLoss()
        crit.__dict__.update(bce=torch.nn.BCEWithLogitsLoss(reduction="none"), stride=torch.tensor([8.0, 16.0, 32.0]), nc=3, no=67, reg_max=16,
                             use_dfl=True, assigner=T.TaskAlignedAssigner(), bbox_loss=bbox, proj=torch.arange(16.0))
        m.criterion = crit
        buf = io.BytesIO()
        torch.save({"epoch": 0, "ema": None, "model": m, "train_args": {"box": 7.5, "cls": 0.5, "dfl": 1.5}}, buf)
    finally:
        for k in fake:
            sys.modules.pop(k, None)
    model, ckpt = attempt_load_one_weight(io.BytesIO(buf.getvalue()))
    assert "ultralytics.utils.loss" not in sys.modules
    assert getattr(model, "criterion", None) is None                 # dropped: first training loss call builds sy11's own
    out = io.BytesIO()
    assert save_checkpoint(out, ema_model=model, updates=1) > 1000    # and the model pickles again (module-level holder classes)
