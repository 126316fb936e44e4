"""GPU: a REFERENCE-format checkpoint (tests/golden/ref_ckpt_t.pt.gz) loaded through sy11.engine.checkpoint runs on the
HIP path and reproduces the oracle's eval output for the same weights; a trainer's checkpoint reloads and predicts the same."""
import gzip
import io

import pytest
import torch

from oracle import yolo11_ref as R
from tests._golden import GOLD

pytestmark = pytest.mark.gpu
DEV = "cuda"


def test_reference_checkpoint_runs_and_matches_oracle():
    from sy11.engine.checkpoint import attempt_load_one_weight
    model, ckpt = attempt_load_one_weight(io.BytesIO(gzip.open(GOLD / "ref_ckpt_t.pt.gz", "rb").read()), device=DEV)
    img = R.seeded_image((2, 3, 64, 64), seed=9)
    with torch.no_grad():
        y, maps = model(img.to(DEV))
    layers = R.resolve_graph("t", nc=2)
    sd = {k: (v.half().float() if v.dtype.is_floating_point and ".dfl." not in k else v)
          for k, v in R.pattern_state_dict(R.empty_state_dict(layers)).items()}
    with torch.no_grad():
        oy, omaps = R.forward(sd, layers, img, train=False)
    scale = float(oy.abs().max())
    assert float((y.cpu() - oy).abs().max()) <= 1e-3 * scale, (float((y.cpu() - oy).abs().max()), scale)
    for a, b in zip(maps, omaps):
        assert float((a.cpu() - b).abs().max()) <= 1e-3 * max(float(b.abs().max()), 1.0)


def test_trainer_checkpoint_round_trip_predicts_identically():
    from sy11.engine.checkpoint import attempt_load_one_weight, save_checkpoint
    from sy11.engine.trainer import DetectionTrainer
    from sy11.nn.tasks import DetectionModel
    m = DetectionModel("yolo11n.yaml", nc=4, verbose=False)
    t = DetectionTrainer(m, batch_size=2, device=DEV, overrides={"amp": False, "nbs": 2}, graphs=False)
    g = torch.Generator().manual_seed(0)
    b = {"img": torch.rand(2, 3, 64, 64, generator=g).to(DEV), "batch_idx": torch.tensor([0., 1.]).to(DEV),
         "cls": torch.tensor([[1.], [2.]]).to(DEV), "bboxes": torch.tensor([[0.5, 0.5, 0.4, 0.4], [0.4, 0.6, 0.3, 0.3]]).to(DEV)}
    for _ in range(2):
        t.train_step(dict(b))
    buf = io.BytesIO()
    save_checkpoint(buf, trainer=t, epoch=1)
    model, ckpt = attempt_load_one_weight(io.BytesIO(buf.getvalue()), device=DEV)
    assert ckpt["updates"] == 2 and ckpt["train_args"]["box"] == 7.5
    ema = t.ema.ema.eval()
    with torch.no_grad():
        y0, _ = ema(b["img"])
        y1, _ = model(b["img"])
    # the checkpoint stores the EMA in f16 (as the reference does): compare against the f16-rounded EMA weights
    assert float((y0 - y1).abs().max()) <= 2e-2 * float(y0.abs().max())
