"""ORACLE — TEST INFRASTRUCTURE ONLY.  A reference-format checkpoint fixture.

Builds the REFERENCE's DetectionModel (scale 't' = [0.5, 0.125, 1024], nc = 2), fills it with the low-entropy pattern
oracle.yolo11_ref.pattern_state_dict (so the gzip'd file stays small), and writes the dictionary trainer.save_model
writes (engine/trainer.py:512-543): pickled ultralytics module objects in 'ema', f16.  Run: python -m oracle.gen_golden_ckpt
-> tests/golden/ref_ckpt_t.pt.gz
"""
from __future__ import annotations

import gzip
import io
import sys
from copy import deepcopy

import torch

from oracle.gen_golden import OUT, ROOT, import_reference


def main():
    import_reference()
    sys.path.insert(0, str(ROOT))
    from oracle.yolo11_ref import pattern_state_dict
    from ultralytics.nn.tasks import DetectionModel, yaml_model_load
    from ultralytics.utils import IterableSimpleNamespace
    d = yaml_model_load("yolo11n.yaml")
    d["scales"]["t"] = [0.5, 0.125, 1024]
    d["scale"] = "t"
    model = DetectionModel(d, ch=3, nc=2, verbose=False)
    model.args = IterableSimpleNamespace(box=7.5, cls=0.5, dfl=1.5, imgsz=64, task="detect")
    model.names = {0: "lte", 1: "nr"}
    model.load_state_dict(pattern_state_dict(model.state_dict()))
    ckpt = {"epoch": 3, "best_fitness": 0.25, "model": None, "ema": deepcopy(model).half(), "updates": 17, "optimizer": None,
            "train_args": {"box": 7.5, "cls": 0.5, "dfl": 1.5, "imgsz": 64, "task": "detect", "data": "synthetic.yaml"},
            "train_metrics": {"fitness": 0.25}, "train_results": {}, "date": "2025-01-01T00:00:00", "version": "8.3.70",
            "license": "AGPL-3.0 (https://ultralytics.com/license)", "docs": "https://docs.ultralytics.com"}
    buf = io.BytesIO()
    torch.save(ckpt, buf)
    with gzip.open(OUT / "ref_ckpt_t.pt.gz", "wb", compresslevel=9) as f:
        f.write(buf.getvalue())
    print("ref_ckpt_t.pt.gz", (OUT / "ref_ckpt_t.pt.gz").stat().st_size, "raw", len(buf.getvalue()))


if __name__ == "__main__":
    main()
