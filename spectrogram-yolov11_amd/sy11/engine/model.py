"""The reference's front door (`from ultralytics import YOLO`: engine/model.py Model :23-1150, models/yolo/model.py YOLO
:12-50) for the detection path this build carries: ``YOLO("yolo11s.yaml" | "weights.pt")`` with ``train`` / ``val`` /
``predict`` / ``__call__`` / ``load`` / ``save`` / ``fuse`` / ``info``, wired to the sy11 trainer, validator and
predictor.  Only what those entry points need of the reference's configuration machinery is here: the dataset YAML rules of
``check_det_dataset`` (data/utils.py:300-373: `train` / `val` required, `names` or `nc`, paths relative to `path` or to
the YAML's folder) and the keyword overrides that map onto this build's trainer arguments.  Everything else the
reference's ``Model`` offers (export, tune, track, benchmark, HUB, callbacks) is out of scope."""
from __future__ import annotations

import os

import random
from pathlib import Path
from types import SimpleNamespace

import numpy as np
import torch

from ..nn.tasks import DetectionModel
from .checkpoint import attempt_load_one_weight, save_checkpoint

_TRAIN_KEYS = {"lr0", "momentum", "weight_decay", "nbs", "box", "cls", "dfl", "amp", "optimizer", "warmup_epochs", "warmup_momentum",
               "warmup_bias_lr", "multi_scale", "imgsz", "deterministic"}


def check_det_dataset(data):
    """data/utils.py:300-373 without downloads: a dict or a YAML path -> dict with absolute `train` / `val`, `names`, `nc`."""
    if not isinstance(data, dict):
        import yaml
        yaml_file = Path(data)
        with open(yaml_file, errors="ignore", encoding="utf-8") as f:
            data = yaml.safe_load(f) or {}
        data["yaml_file"] = str(yaml_file)
    data = dict(data)
    for k in ("train", "val"):
        if k not in data:
            if k != "val" or "validation" not in data:
                raise SyntaxError(f"'{k}:' key missing. 'train' and 'val' are required in all data YAMLs.")
            data["val"] = data.pop("validation")
    if "names" not in data and "nc" not in data:
        raise SyntaxError("either 'names' or 'nc' are required in all data YAMLs.")
    if "names" in data and "nc" in data and len(data["names"]) != data["nc"]:
        raise SyntaxError(f"'names' length {len(data['names'])} and 'nc: {data['nc']}' must match.")
    if "names" not in data:
        data["names"] = [f"class_{i}" for i in range(data["nc"])]
    else:
        data["nc"] = len(data["names"])
    if isinstance(data["names"], (list, tuple)):
        data["names"] = dict(enumerate(data["names"]))
    data["names"] = {int(k): str(v) for k, v in data["names"].items()}
    yaml_dir = Path(data["yaml_file"]).resolve().parent if data.get("yaml_file") else Path.cwd()
    root = Path(data.get("path") or yaml_dir)
    if not root.is_absolute():                               # the reference anchors a relative `path` at its global datasets dir;
        root = (yaml_dir / root).resolve()                   # without that setting the YAML's own folder is the only sensible anchor
    data["path"] = root
    for k in ("train", "val", "test"):
        if data.get(k):
            x = data[k]
            data[k] = str((root / x).resolve()) if isinstance(x, str) else [str((root / v).resolve()) for v in x]
    return data


class YOLO:
    """models/yolo/model.py:12 / engine/model.py:23 for task="detect"."""

    def __init__(self, model="yolo11n.yaml", task="detect", verbose=False, device="cuda", nc=None):
        if task not in (None, "detect"):
            raise NotImplementedError("sy11 carries the detection task only")
        self.task, self.device = "detect", torch.device(device)
        self.ckpt, self.trainer, self.predictor, self.metrics = None, None, None, None
        self.overrides = {}
        name = str(model)
        if name.endswith((".yaml", ".yml")):
            self.model = DetectionModel(name, nc=nc, verbose=verbose) if nc else DetectionModel(name, verbose=verbose)
            self.cfg = name
        else:
            self.model, self.ckpt = attempt_load_one_weight(name, device=self.device)
            self.cfg = getattr(self.model, "yaml_file", None)
        self.model_name = name

    # ---- bookkeeping of the reference's Model
    @property
    def names(self):
        return getattr(self.model, "names", None)

    def info(self, detailed=False, verbose=True):
        return self.model.info(detailed=detailed, verbose=verbose)

    def fuse(self):
        self.model.fuse()
        return self

    def load(self, weights):
        """engine/model.py:326-350 — take matching weights from a checkpoint / state_dict (class count may differ)."""
        src = attempt_load_one_weight(weights, device=self.device)[0].state_dict() if isinstance(weights, (str, Path)) else weights
        own = self.model.state_dict()
        ok = {k: v for k, v in src.items() if k in own and own[k].shape == v.shape}
        self.model.load_state_dict(ok, strict=False)
        return self

    def save(self, filename="saved_model.pt"):
        save_checkpoint(filename, ema_model=self.model, updates=0)
        return self

    def _nc_model(self, nc, names):
        """engine/trainer.py get_model: the YAML is re-instantiated with the dataset's class count; weights carry over."""
        if getattr(self.model.model[-1], "nc", nc) != nc:
            fresh = DetectionModel(self.cfg or self.model.yaml, nc=nc, verbose=False)
            own = fresh.state_dict()
            fresh.load_state_dict({k: v for k, v in self.model.state_dict().items() if k in own and own[k].shape == v.shape}, strict=False)
            self.model = fresh
        self.model.names = names
        self.model.nc = nc
        return self.model

    # ---- train / val / predict
    def train(self, data, epochs=100, batch=16, imgsz=640, workers=8, seed=0, save_dir=None, close_mosaic=10, patience=100, lrf=0.01,
              cos_lr=False, resume=False, val=True, **overrides):
        """engine/model.py:754-840 -> DetectionTrainer (engine/trainer.py): one process per GPU; under torchrun the RANK /
        WORLD_SIZE of the environment select the shard and switch on the RCCL gradient sum."""
        from ..data.dataset import build_dataloader, build_yolo_dataset
        from . import ddp
        from .trainer import DetectionTrainer
        devices = _device_list(overrides.pop("device", None))
        if len(devices) > 1 and "RANK" not in os.environ:
            # engine/trainer.py:170-207 + utils/dist.py:25-66: `device=[0, 1, ...]` outside a launcher -> one CHILD process per
            # GPU runs this very call; the parent waits and then continues with best.pt, as the reference does
            return self._train_ddp(devices, dict(data=data, epochs=epochs, batch=batch, imgsz=imgsz, workers=workers, seed=seed,
                                                 save_dir=save_dir, close_mosaic=close_mosaic, patience=patience, lrf=lrf, cos_lr=cos_lr,
                                                 resume=resume, val=val, **overrides))
        if len(devices) == 1 and "RANK" not in os.environ:
            self.device = torch.device("cuda", devices[0])
        unknown = set(overrides) - _TRAIN_KEYS - set(_hyp_defaults())
        if unknown:
            raise SyntaxError(f"unknown train arguments {sorted(unknown)} (sy11 carries {sorted(_TRAIN_KEYS | set(_hyp_defaults()))})")
        data = check_det_dataset(data)
        rank, local, world = ddp.setup_process_group()
        dev = torch.device("cuda", local) if self.device.type == "cuda" else self.device
        torch.manual_seed(seed + 1 + rank); random.seed(seed + 1 + rank); np.random.seed(seed + 1 + rank)      # trainer.py:107
        model = self._nc_model(data["nc"], data["names"])
        tr_over = {k: v for k, v in overrides.items() if k in _TRAIN_KEYS}
        tr_over["imgsz"] = imgsz
        self.trainer = DetectionTrainer(model, batch_size=batch, device=dev, overrides=tr_over, world_size=world)
        if resume and self.ckpt is not None:
            start = self.trainer.resume_training(self.ckpt)
        else:
            start = 0
        hyp = SimpleNamespace(**{**_hyp_defaults(), **{k: v for k, v in overrides.items() if k in _hyp_defaults()}}, imgsz=imgsz)
        stride = int(max(model.stride))
        ds = build_yolo_dataset(hyp, data["train"], batch, data, mode="train", stride=stride, device=dev)
        dl = build_dataloader(ds, batch, workers=workers, shuffle=True, rank=rank if world > 1 else -1, world_size=world,
                              out=self.trainer.batch_buffer(imgsz), dtype=torch.float32)
        val_batches = None
        if val and data.get("val"):
            vds = build_yolo_dataset(hyp, data["val"], batch * 2, data, mode="val", rect=True, stride=stride, device=dev)
            vdl = build_dataloader(vds, batch * 2, workers=workers, shuffle=False)
            val_batches = lambda: vdl                                                                            # noqa: E731
        save_dir = save_dir or Path("runs") / "detect" / "train"
        hist = self.trainer.fit(dl, epochs, val_batches=val_batches, save_dir=save_dir, close_mosaic=close_mosaic, start_epoch=start,
                                lrf=lrf, cos_lr=cos_lr, patience=patience)
        self.metrics = hist[-1]["metrics"] if hist else None
        if rank in (-1, 0):                                                      # the per-epoch records (what results.csv holds in the
            import json                                                          # reference); a launching parent reads them back
            Path(save_dir).mkdir(parents=True, exist_ok=True)
            (Path(save_dir) / "results.json").write_text(json.dumps(hist, default=str))
        best = Path(save_dir) / "best.pt"
        if rank in (-1, 0) and best.exists():                                    # engine/model.py:833-837: continue with best.pt
            self.model, self.ckpt = attempt_load_one_weight(str(best), device=dev)
        return hist

    def _train_ddp(self, devices, kw):
        """Spawn len(devices) ranks of `YOLO(<same model>).train(<same arguments>)` and wait (sy11.engine.ddp.launch)."""
        import json
        import tempfile
        from . import ddp
        from .. import _lib
        if torch.cuda.is_initialized():
            raise _lib.Sy11Error("train(device=[...]) starts one process per GPU and must do so before THIS process has touched "
                                 "the GPU (construct YOLO(...) from a .yaml, or with device='cpu'); alternatively launch the "
                                 "script with `python -m torch.distributed.run --nproc-per-node N`")
        save_dir = str(kw.get("save_dir") or Path("runs") / "detect" / "train")
        kw = {**kw, "save_dir": save_dir, "data": kw["data"] if isinstance(kw["data"], dict) else str(kw["data"])}
        payload = {"pkg": str(Path(__file__).resolve().parents[2]), "model": self.model_name, "nc": getattr(self.model.model[-1], "nc", None),
                   "kw": kw}
        # guarded: anything that re-imports this file as a module (multiprocessing's spawn bootstrap does) must not train again
        src = ("import json, sys\n"
               "if __name__ == '__main__':\n"
               f"    P = json.loads({json.dumps(json.dumps(payload, default=str))})\n"
               "    sys.path.insert(0, P['pkg'])\n"
               "    from sy11.engine.model import YOLO\n"
               "    m = YOLO(P['model'], nc=P['nc']) if str(P['model']).endswith(('.yaml', '.yml')) else YOLO(P['model'])\n"
               "    m.train(**P['kw'])\n")
        with tempfile.NamedTemporaryFile("w", suffix=".py", prefix="_sy11_ddp_", delete=False) as f:
            f.write(src)
        vis = ",".join(str(d) for d in devices)
        env = {**os.environ, "HIP_VISIBLE_DEVICES": vis, "CUDA_VISIBLE_DEVICES": vis}
        if "SY11_FORCE_DEVICE" in os.environ:                  # rehearsal of N ranks on fewer GPUs (ddp.setup_process_group): the ranks
            env = dict(os.environ)                             # share the forced device, the visibility list is left alone
        try:
            ddp.launch([f.name], len(devices), env=env)
        finally:
            os.unlink(f.name)
        best = Path(save_dir) / "best.pt"
        last = Path(save_dir) / "last.pt"
        ck = best if best.exists() else last
        if ck.exists():
            self.model, self.ckpt = attempt_load_one_weight(str(ck), device="cpu")
        rec = Path(save_dir) / "results.json"
        hist = json.loads(rec.read_text()) if rec.exists() else []               # same return type as the single-process branch
        self.metrics = hist[-1]["metrics"] if hist else None
        return hist

    def val(self, data=None, batch=32, imgsz=640, conf=0.001, iou=0.7, half=False, workers=8, **kw):
        """engine/model.py:623-670 -> DetectionValidator over the rect val loader."""
        from ..data.dataset import build_dataloader, build_yolo_dataset
        from .validator import DetectionValidator
        data = check_det_dataset(data)
        hyp = SimpleNamespace(**_hyp_defaults(), imgsz=imgsz)
        self.model.names = data["names"]
        vds = build_yolo_dataset(hyp, data["val"], batch, data, mode="val", rect=True, stride=int(max(self.model.stride)), device=self.device)
        vdl = build_dataloader(vds, batch, workers=workers, shuffle=False)
        self.metrics = DetectionValidator(self.model, device=self.device, conf=conf, iou=iou, half=half)(self.model, vdl)
        return self.metrics

    def predict(self, source, conf=0.25, iou=0.7, imgsz=640, max_det=300, classes=None, agnostic_nms=False, half=False, **kw):
        """engine/model.py:500-560 -> DetectionPredictor.  ``source``: an (h, w, 3) BGR uint8 array, a list of them, image / .npy
        file paths, or a (B, 3, H, W) tensor."""
        from ..data.dataset import read_image
        from .predictor import DetectionPredictor
        if self.predictor is None or self.predictor.args["conf"] != conf or self.predictor.args["iou"] != iou or self.predictor.imgsz != ((imgsz, imgsz) if isinstance(imgsz, int) else tuple(imgsz)):
            self.predictor = DetectionPredictor(self.model, device=self.device, conf=conf, iou=iou, max_det=max_det, classes=classes,
                                                agnostic_nms=agnostic_nms, half=half, imgsz=imgsz)
        paths = None
        if isinstance(source, (str, Path)):
            source = [source]
        if isinstance(source, np.ndarray):
            source = [source]
        if isinstance(source, (list, tuple)):
            paths = [str(s) if isinstance(s, (str, Path)) else None for s in source]
            source = [read_image(s) if isinstance(s, (str, Path)) else s for s in source]
        return self.predictor(source, paths=paths)

    __call__ = predict


def _device_list(device):
    """`device=0`, `"0,1"`, `[0, 1]`, `"cuda:1"` -> list of GPU indices (utils/torch_utils.py select_device's parsing)."""
    if device is None or device == "" or str(device) == "cpu":
        return []
    from .. import _lib
    try:
        if isinstance(device, (list, tuple)):
            return [int(d) for d in device]
        txt = str(device).lower().replace("cuda:", "").replace("(", "").replace(")", "").replace("[", "").replace("]", "").replace(" ", "")
        if txt == "cuda":
            return [0]
        return [int(d) for d in txt.split(",") if d != ""]
    except (TypeError, ValueError):
        raise _lib.Sy11Error(f"device={device!r}: expected 'cpu', a GPU index, 'cuda:N', '0,1' or a list of indices "
                             f"(the MI355X is the only accelerator this build drives)") from None


def _hyp_defaults():
    from ..data.dataset import DEFAULT_HYP
    return dict(DEFAULT_HYP)
