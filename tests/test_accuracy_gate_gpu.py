"""GPU: the accuracy gate of BASELINE configs[4] / north_star ("mAP@0.5 within 0.1 of reference on the held-out set") on a
synthetic held-out set generated from IQ (oracle/synth_iq.py: bursts and chirps with known time-frequency boxes; the reference's
dataset is not distributed).  tools/accuracy_gate.py does the work; here at test size:
  * scale-t model: the HIP f32 trainer and the ORACLE trainer (CPU restatement, autograd backward) start from the same weights and
    see the same mini-batches — loss curve within 1e-2 over the first SGD steps, mAP@0.5 after the AdamW schedule within 0.1;
  * yolo11n: HIP f16 (AMP, the configs[4] dtype) against HIP f32, mAP@0.5 within 0.1.
The full-size runs (more steps, the fusion variant) are recorded in profiles/r02/accuracy_gate*.json."""
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tools"))


def test_hip_trainer_and_oracle_trainer_reach_the_same_map_on_held_out_iq_scenes():
    import accuracy_gate as A
    # 320 steps: near the plateau (at 160 steps mAP still rises ~0.1 per 40 steps); 64 held-out scenes and the mean of two HIP
    # runs: HIP training is not bit-reproducible and one run on 24 scenes spreads by +-0.05 on its own
    r = A.run("tiny", steps=320, batch=8, n_train=64, n_val=64, curve_steps=10, repeats=2)
    assert r["gate"]["loss_curve_max_rel_dev_first_steps"] <= 1e-2, (r["curve_sgd"], r["gate"])
    assert r["oracle"]["map50"] > 0.3 and r["hip_f32"]["map50"] > 0.3, (r["oracle"]["map50"], r["hip_f32"]["map50"])      # both learned
    assert r["gate"]["map50_abs_diff_f32_vs_oracle"] <= 0.1, r["gate"]


def test_f16_and_f32_training_reach_the_same_map():
    import accuracy_gate as A
    r = A.run("n", steps=200, oracle_steps=1, batch=16, n_train=64, n_val=32, curve_steps=1)
    assert r["hip_f32"]["map50"] > 0.5 and r["hip_f16"]["map50"] > 0.5, (r["hip_f32"]["map50"], r["hip_f16"]["map50"])
    assert r["gate"]["map50_abs_diff_f16_vs_f32"] <= 0.1, r["gate"]
