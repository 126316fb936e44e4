#!/usr/bin/env python3
"""bench.py — spectrogram-images/s of the hot path on MI355X (contract: see the round prompt / DESIGN.md §Measurement).

A "step" = one pass of the hot path over one synthetic batch that is ALREADY RESIDENT IN HBM:
  IQ (B, 164608) complex64 -> HIP STFT/log-mel -> (B,3,640,640) -> YOLOv11-s forward (train-mode BN) -> v8 loss
  -> backward (dgrad/wgrad/BN) -> [RCCL gradient sum when N > 1] -> unscale, clip 10, SGD-nesterov step, EMA.
One process per GPU; for N > 1 launch with torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE from the env).
Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT / "spectrogram-yolov11_amd"))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

FWD_GFLOP_PER_IMG = 21.467          # 88 Conv2d of yolo11s @ 640x640 (algorithmic, 2*MAC)
TRAIN_GFLOP_PER_IMG = 64.40         # fwd + dgrad + wgrad (first layer has no dgrad)
PEAK_TFLOPS = {"f16": 2500.0, "bf16": 2500.0, "f32": 157.3}   # dense MFMA peaks, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0               # HBM3E spec peak (~6300 achievable), MI355X_MICROARCH.md


def synthetic_iq(batch, n_samples, seed, device):
    """complex white noise + an OFDM-like band-limited burst + a chirp, generated on the device."""
    g = torch.Generator(device=device).manual_seed(seed)
    t = torch.arange(n_samples, device=device, dtype=torch.float32)
    noise = torch.randn(batch, n_samples, 2, generator=g, device=device) * (0.1 / math.sqrt(2))
    iq = torch.view_as_complex(noise).clone()
    f0 = -0.25 + 0.5 * torch.rand(batch, 1, generator=g, device=device)
    k = torch.arange(16, device=device, dtype=torch.float32).view(1, 16, 1)
    ph = torch.rand(batch, 16, 1, generator=g, device=device) * 2 * math.pi
    t0, t1 = int(n_samples * 0.2), int(n_samples * 0.55)
    seg = t[t0:t1].view(1, 1, -1)
    burst = torch.exp(1j * (2 * math.pi * (f0.view(-1, 1, 1) + (k - 8) * 0.005) * seg + ph)).sum(1) / 4.0
    iq[:, t0:t1] += burst.to(torch.complex64)
    tt = t[int(n_samples * 0.6):]
    tt = tt - tt[0]
    phase = 2 * math.pi * (0.1 * tt + 0.5 * 0.25 / tt.numel() * tt * tt)
    iq[:, int(n_samples * 0.6):] += (0.7 * torch.exp(1j * phase)).to(torch.complex64)
    return iq.contiguous()


def synthetic_labels(batch, seed, device, nc=80):
    g = torch.Generator().manual_seed(seed)
    n = batch * 3
    cls = torch.randint(0, nc, (n, 1), generator=g).float()
    cxcy = 0.25 + 0.5 * torch.rand(n, 2, generator=g)
    wh = 0.05 + 0.3 * torch.rand(n, 2, generator=g)
    return {"batch_idx": torch.arange(batch).repeat_interleave(3).float().to(device), "cls": cls.to(device),
            "bboxes": torch.cat((cxcy, wh), 1).to(device)}


def cpu_baseline(batch=4, imgsz=640, iters=2):
    """The oracle (CPU restatement of the reference path, plain PyTorch fp32) timed on this host's cores."""
    sys.path.insert(0, str(ROOT))
    from oracle import loss_ref, yolo11_ref as R
    torch.manual_seed(0)
    layers = R.resolve_graph("s", nc=80)
    sd = R.seeded_state_dict(R.empty_state_dict(layers), seed=0)
    for k, v in sd.items():
        if v.dtype.is_floating_point and "running" not in k:
            v.requires_grad_(True)
    img = torch.rand(batch, 3, imgsz, imgsz)
    lab = synthetic_labels(batch, 0, "cpu")
    times = []
    for it in range(iters + 1):
        t0 = time.perf_counter()
        maps = R.forward(sd, layers, img, train=True)
        loss, _ = loss_ref.detection_loss(maps, lab, nc=80)
        loss.backward()
        for v in sd.values():
            v.grad = None
        if it:
            times.append(time.perf_counter() - t0)
    dt = sum(times) / len(times)
    return {"value": round(batch / dt, 3), "unit": "spectrogram-images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{iters} timed + 1 warm-up train fwd+loss+bwd iterations of yolo11s {imgsz}x{imgsz} at batch {batch} "
                      f"(oracle/yolo11_ref.py + loss_ref.py, fp32, torch {torch.get_num_threads()} threads)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU")
    ap.add_argument("--imgsz", type=int, default=640)
    ap.add_argument("--model", default="yolo11s.yaml")
    ap.add_argument("--dtype", default="f16", choices=["f16", "f32"])
    ap.add_argument("--mode", default="train", choices=["train", "fwd"])
    ap.add_argument("--no-stft", action="store_true", help="feed resident (B,3,H,W) images instead of IQ")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-fwd-leg", action="store_true", help="skip the forward-only (configs[1]) leg reported next to the headline")
    ap.add_argument("--no-graphs", action="store_true", help="launch every kernel individually instead of hipGraph replay")
    a = ap.parse_args()

    from sy11 import _lib
    from sy11.data.spectrogram import SpectrogramProducer
    from sy11.engine import ddp
    from sy11.engine.trainer import DetectionTrainer
    from sy11.nn.tasks import DetectionModel

    rank, local, world = ddp.setup_process_group(os.environ.get("SY11_DDP_BACKEND"))
    if "SY11_FORCE_DEVICE" in os.environ:                  # rehearsal of the N>1 path on a one-GPU box (gloo backend)
        local = int(os.environ["SY11_FORCE_DEVICE"])
    if world != a.gpus and a.gpus > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    torch.manual_seed(0 + 1 + rank)                        # trainer.py:107 init_seeds(seed + 1 + RANK)

    model = DetectionModel(a.model, nc=80, verbose=False)
    producer = SpectrogramProducer(dev, n_frames=a.imgsz, n_mel=a.imgsz) if not a.no_stft else None
    tr = DetectionTrainer(model, batch_size=a.batch, device=dev, overrides={"amp": a.dtype == "f16"}, world_size=world,
                          producer=producer, graphs=not a.no_graphs)
    labels = synthetic_labels(a.batch, 100 + rank, dev)
    if producer is not None:
        data = {"iq": synthetic_iq(a.batch, producer.n_samples, 1 + rank, dev)}
    else:
        data = {"img": torch.rand(a.batch, 3, a.imgsz, a.imgsz, device=dev)}

    def step():
        batch = {**data, **labels}
        if a.mode == "train":
            return tr.train_step(batch)
        with torch.no_grad():
            tr.model.train()
            return tr.model(tr.preprocess_batch(batch)["img"])

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = tmax.item()
    ms = dt / a.steps * 1e3
    value = a.batch * world * a.steps / dt

    fwd_only = None
    if rank == 0 and world == 1 and a.mode == "train" and not a.no_fwd_leg:
        # configs[1] of BASELINE.json, reported next to the headline (never part of `value`): forward only, batch statistics,
        # resident spectrogram tensors, the same captured graph machinery
        img = tr.preprocess_batch({**data, **labels})["img"].clone()
        tr.model.train()

        def fwd():
            with torch.no_grad():
                return tr.model(img)
        for _ in range(3):
            fwd()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(a.steps):
            fwd()
        torch.cuda.synchronize()
        fdt = (time.perf_counter() - t1) / a.steps
        fwd_only = {"workload": "configs[1]: YOLOv11-s forward only, bs=64, resident spectrogram tensors", "value": round(a.batch / fdt, 1),
                    "unit": "spectrogram-images/s", "ms_per_step": round(fdt * 1e3, 3),
                    "conv_tflops": round(a.batch / fdt * FWD_GFLOP_PER_IMG / 1e3, 2),
                    "conv_roofline_frac": round(a.batch / fdt * FWD_GFLOP_PER_IMG / 1e3 / PEAK_TFLOPS[a.dtype], 4)}

    roof = None
    if world > 1 and not a.no_roofline:
        step()                                                  # every rank takes the extra step: its gradient all-reduce must be matched
    if rank == 0 and not a.no_roofline:
        # one instrumented step: every C-ABI launch bracketed by events on the launch stream
        tr.model.__dict__.pop("_sy11_graph_cfg", None)          # per-launch events need individually launched kernels
        _lib.PROFILE = []
        if world == 1:
            step()
        else:                                                   # the collective already happened above: instrument forward+backward only
            store = tr.model.__dict__.get("_sy11_grads")
            from sy11.engine import module_post_backward
            hook = module_post_backward.pop(id(store), None) if store is not None else None
            step()
            if hook is not None:
                module_post_backward[id(store)] = hook
        torch.cuda.synchronize()
        prof, _lib.PROFILE = _lib.PROFILE, None
        fam = {}
        for name, e0, e1, meta in prof:
            f = fam.setdefault(name, {"ms": 0.0, "n": 0, "flops": 0.0, "bytes": 0.0})
            f["ms"] += e0.elapsed_time(e1)
            f["n"] += 1
            if meta and (name.startswith("sy11_conv2d") or name.startswith("sy11_stem")):
                f["flops"] += meta["flops"]
                f["bytes"] += meta["bytes"]
        dump = os.environ.get("SY11_DUMP_LAUNCHES")
        if dump:
            with open(dump, "w") as fh:
                for name, e0, e1, meta in prof:
                    fh.write(json.dumps({"name": name, "ms": round(e0.elapsed_time(e1), 4),
                                         "gflop": round((meta or {}).get("flops", 0) / 1e9, 3) if name.startswith(("sy11_conv2d", "sy11_stem")) else 0,
                                         "mbytes": round((meta or {}).get("bytes", 0) / 1e6, 3) if name.startswith(("sy11_conv2d", "sy11_stem")) else 0,
                                         "desc": (meta or {}).get("desc") if name.startswith(("sy11_conv2d", "sy11_stem")) else None}) + "\n")
        top = max(fam.items(), key=lambda kv: kv[1]["ms"])
        name, f = top
        peak = PEAK_TFLOPS[a.dtype]
        ach = f["flops"] / (f["ms"] * 1e-3) / 1e12 if f["ms"] > 0 else 0.0
        traffic = None          # HBM bytes per launch from rocprofv3 PMC passes (cannot be collected from inside this process)
        tj = ROOT / "profiles" / "r01" / "traffic.json"
        if tj.exists():
            fam_t = json.loads(tj.read_text()).get("families", {}).get(name)
            if fam_t:
                traffic = round(fam_t["bytes_per_launch"])
        # which roof binds the family: its arithmetic intensity (algorithmic FLOPs / algorithmic bytes) against the ridge
        # peak_flops / peak_bandwidth.  yolo11s' conv families sit at ~140 FLOP/B in f16, below the 312 FLOP/B ridge -> HBM.
        secs = f["ms"] * 1e-3
        ai = f["flops"] / max(f["bytes"], 1.0)
        if f["bytes"] > 0 and ai < peak * 1e12 / (PEAK_HBM_GBS * 1e9):
            ach_b = f["bytes"] / secs / 1e9 if secs > 0 else 0.0
            roof = {"kernel": name, "bound": "hbm", "achieved": round(ach_b, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": round(ach_b / PEAK_HBM_GBS, 4)}
        else:
            roof = {"kernel": name, "bound": "mfma", "achieved": round(ach, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(ach / peak, 4)}
        roof.update({"traffic": traffic, "arithmetic_intensity_flop_per_byte": round(ai, 1),
                     "mfma_tflops": round(ach, 2), "mfma_frac": round(ach / peak, 4),
                "algorithmic_bytes_per_launch": round(f["bytes"] / max(f["n"], 1)), "launches": f["n"],
                "avg_launch_ms": round(f["ms"] / max(f["n"], 1), 4),
                "families_ms": {k: round(v["ms"], 3) for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms"])}})

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline()

    if fwd_only is not None:
        # the deployed forward: BatchNorm folded into the convs (model.fuse()), eval mode, Detect decode included.  Last thing
        # this process does with the model (fusing is destructive); reported inside `forward_only`, never part of `value`.
        from sy11.engine import enable_graphs
        m = tr.model
        enable_graphs(m)                                        # fresh entries: the eval forward captures its own graph
        m.eval()
        m.fuse()

        def pred():
            with torch.no_grad():
                return m(img)
        for _ in range(3):
            pred()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        for _ in range(a.steps):
            pred()
        torch.cuda.synchronize()
        pdt = (time.perf_counter() - t2) / a.steps
        fwd_only["fused_eval"] = {"workload": "model.fuse() + eval forward incl. Detect decode (predictor path), bs=64", "value": round(a.batch / pdt, 1),
                                  "ms_per_step": round(pdt * 1e3, 3), "conv_tflops": round(a.batch / pdt * FWD_GFLOP_PER_IMG / 1e3, 2),
                                  "conv_roofline_frac": round(a.batch / pdt * FWD_GFLOP_PER_IMG / 1e3 / PEAK_TFLOPS[a.dtype], 4)}

    if rank == 0:
        gflop = TRAIN_GFLOP_PER_IMG if a.mode == "train" else FWD_GFLOP_PER_IMG
        out = {
            "metric": "spectrogram-images/sec (train fwd+bwd) YOLOv11-s 640², bs=64, 1/2/4/8 GPU" if a.mode == "train"
            else "spectrogram-images/sec (forward only) YOLOv11-s 640², bs=64",
            "value": round(value, 2), "unit": "spectrogram-images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": a.dtype, "data": "synthetic",
            "config": {"workload": ("configs[2]: YOLOv11-s train fwd+bwd (+clip, SGD-nesterov, EMA) from synthetic IQ through the "
                                    "HIP STFT/log-mel producer" if a.mode == "train" and not a.no_stft else
                                    "configs[1]: YOLOv11-s forward" if a.mode == "fwd" else
                                    "YOLOv11-s train fwd+bwd from resident spectrogram tensors"),
                       "model": a.model, "imgsz": a.imgsz, "batch_per_gpu": a.batch, "global_batch": a.batch * world,
                       "parallelism": f"dp{world}", "weights": "random-init", "nc": 80},
            "conv_tflops": round(value * gflop / 1e3, 2),
            "conv_roofline_frac": round(value * gflop / 1e3 / PEAK_TFLOPS[a.dtype], 4),
            "roofline": roof, "cpu_baseline": cpu, "forward_only": fwd_only,
        }
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
