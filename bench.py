#!/usr/bin/env python3
"""bench.py — spectrogram-images/s of the hot path on MI355X (contract: see the round prompt / DESIGN.md §Measurement).

A "step" = one pass of the hot path over one synthetic batch that is ALREADY RESIDENT IN HBM:
  IQ (B, 164608) complex64 -> HIP STFT/log-mel -> (B,3,640,640) -> YOLOv11-s forward (train-mode BN) -> v8 loss
  -> backward (dgrad/wgrad/BN) -> [RCCL gradient sum when N > 1] -> unscale, clip 10, SGD-nesterov step, EMA.
One process per GPU.  `python3 bench.py --gpus N` spawns its own N ranks (sy11.engine.ddp.launch: fresh child processes, started before
this one touches the GPU); under torch.distributed.run it takes RANK / LOCAL_RANK / WORLD_SIZE from the env instead.
Rank 0 prints ONE JSON line.

Besides the headline (`value`, configs[2]) the line carries, all measured after the timed region and never part of `value`:
  roofline      the kernel family with the largest share of the step (HIP events around every C-ABI launch of one instrumented
                step, issued behind a head-start delay so that the GPU never waits for the host between launches), its algorithmic
                bytes / FLOPs, the HBM traffic of the same family from the rocprofv3 PMC passes (profiles/r04/traffic.json, quoted only when measured on this csrc);
  peaks         measured on this GPU in this run: pure-MFMA loop (sy11_peak_mfma_f16) and a 1 GiB device-to-device copy;
  cpu_baseline  the oracle (CPU restatement of the reference path) on the host cores: 7 threads (the reference's default
                min(8, ncpu - 1)) and all cores, eval forward and train step;
  forward_only  configs[1]; extra: the f32 train line and configs[4]'s model (fusion variant, nc = 2) at the same batch / size.
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
FUSION_FWD_GFLOP_PER_IMG = 18.892   # yolo11s_fusion_sand3_new, nc = 2
PEAK_TFLOPS = {"f16": 2500.0, "bf16": 2500.0, "f32": 157.3}   # dense MFMA peaks, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0               # HBM3E spec peak (~6300 achievable), MI355X_MICROARCH.md
PROFILE_DIR = ROOT / "profiles" / "r04"


def kernel_source_sha16():
    """sha256 (first 16 hex digits) over the kernel sources (csrc/*.hip, *.h, in name order): the committed PMC traffic figures
    are only quoted when they were measured on THIS source (tools/pmc_summary.py records the same hash; the GPU box has no .git)."""
    import hashlib
    h = hashlib.sha256()
    src = ROOT / "spectrogram-yolov11_amd" / "csrc"
    for f in sorted(list(src.glob("*.hip")) + list(src.glob("*.h"))):
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()[:16]

# kernel family -> the C-ABI entry points that launch it, and the kernel symbols a rocprofv3 trace shows for it
FAMILIES = {
    "conv fwd+dgrad (dense)": {"calls": ("sy11_conv2d_fwd", "sy11_conv2d_dgrad"), "symbols": ("igemm_kernel", "igemm8_kernel", "igemm1x1p_kernel", "halo3x3_kernel", "halo_dgrad_s2_kernel", "smallc3x3_kernel")},
    "conv wgrad (dense)": {"calls": ("sy11_conv2d_wgrad",), "symbols": ("wgrad16_kernel", "wgrad16w_kernel", "wgrad3x3p_kernel", "wgrad_kernel")},
    "depthwise conv": {"calls": (), "symbols": ("dw3x3_kernel", "dwconv_")},
    "batchnorm": {"calls": ("sy11_bn_act_fwd", "sy11_bn_act_bwd_reduce", "sy11_bn_act_bwd_apply", "sy11_bn_act_bwd_apply_res", "sy11_bn_finalize"),
                  "symbols": ("bn_act_fwd_kernel", "bn_bwd_reduce_kernel", "bn_bwd_apply_kernel", "bn_finalize_kernel")},
    "stem": {"calls": ("sy11_stem_conv_fwd", "sy11_stem_conv_wgrad"), "symbols": ("stem_fwd_tile", "stem_wgrad_tile", "stem_fwd_mma", "stem_wgrad_mma")},
}


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


def host_cpus():
    """CPUs this process may really use: the affinity mask capped by the cgroup CPU quota (a GPU box hands one GPU's share of
    the host — 16 CPUs — to a job while every core of the machine stays visible)."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(f).read().split()
            if f.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    n = min(n, max(1, q // int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())))
            break
        except Exception:
            continue
    return n


def cpu_baseline(batch=8, imgsz=640, leg_budget_s=12.0, full_batch=64):
    """SURVEY §8(d) / BASELINE.md §3: the oracle (plain PyTorch fp32 restatement of the reference path) on this host's cores, at
    n = 7 threads (the reference's own default min(8, ncpu - 1), utils/__init__.py:44) and n = all usable cores; warm-up + up to 5
    timed iterations each of (a) eval forward and (b) train fwd + loss + bwd, every leg bounded to ~20 s (>= 2 timed iterations).
    A bounded sample of the bench workload: same model, size and inputs at batch 8 (a train iteration at batch 64 takes a minute)."""
    sys.path.insert(0, str(ROOT))
    from oracle import loss_ref, yolo11_ref as R
    t_start = time.perf_counter()
    torch.manual_seed(0)
    layers = R.resolve_graph("s", nc=80)
    sd = R.seeded_state_dict(R.empty_state_dict(layers), seed=0)
    img = torch.rand(batch, 3, imgsz, imgsz)
    lab = synthetic_labels(batch, 0, "cpu")
    ncpu = host_cpus()
    runs = []

    def timed(fn, warm, iters, what):
        t_leg = time.perf_counter()
        for _ in range(warm):
            fn()
        ts = []
        for _ in range(iters):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
            if time.perf_counter() - t_leg > leg_budget_s and len(ts) >= 2:
                break
        print(f"[bench] cpu baseline: {what}: {len(ts)} iterations, {sum(ts) / len(ts):.2f} s each", file=sys.stderr, flush=True)
        return sum(ts) / len(ts), len(ts)

    for n in sorted({min(7, ncpu), ncpu}):
        torch.set_num_threads(n)
        esd = {k: v.clone() for k, v in sd.items()}

        def fwd():
            with torch.no_grad():
                R.forward(esd, layers, img, train=False)
        dt, it = timed(fwd, 1, 3, f"eval forward, {n} threads")
        runs.append({"mode": "eval forward", "threads": n, "batch": batch, "iters": it, "img_s": round(batch / dt, 3),
                     "gflops": round(batch / dt * FWD_GFLOP_PER_IMG, 1)})
        tsd = {k: v.clone() for k, v in sd.items()}
        for k, v in tsd.items():
            if v.dtype.is_floating_point and "running" not in k:
                v.requires_grad_(True)

        def train():
            maps = R.forward(tsd, layers, img, train=True)
            loss, _ = loss_ref.detection_loss(maps, lab, nc=80)
            loss.backward()
            for v in tsd.values():
                v.grad = None
        dt, it = timed(train, 1, 3, f"train fwd+loss+bwd, {n} threads")
        runs.append({"mode": "train fwd+loss+bwd", "threads": n, "batch": batch, "iters": it, "img_s": round(batch / dt, 3),
                     "gflops": round(batch / dt * TRAIN_GFLOP_PER_IMG, 1)})
    # ... and the workload's own batch (64) once on all cores: 1 warm-up + 2 timed train iterations (~25 s of CPU work, ~35 GB of
    # saved activations); falls back to the batch-8 figure when the host cannot hold it
    full = None
    if full_batch and full_batch != batch:
        try:
            torch.set_num_threads(ncpu)
            img64 = torch.rand(full_batch, 3, imgsz, imgsz)
            lab64 = synthetic_labels(full_batch, 0, "cpu")
            tsd = {k: v.clone() for k, v in sd.items()}
            for k, v in tsd.items():
                if v.dtype.is_floating_point and "running" not in k:
                    v.requires_grad_(True)

            def train64():
                maps = R.forward(tsd, layers, img64, train=True)
                loss, _ = loss_ref.detection_loss(maps, lab64, nc=80)
                loss.backward()
                for v in tsd.values():
                    v.grad = None
            dt, it = timed(train64, 1, 2, f"train fwd+loss+bwd at batch {full_batch}, {ncpu} threads")
            full = {"mode": "train fwd+loss+bwd", "threads": ncpu, "batch": full_batch, "iters": it, "img_s": round(full_batch / dt, 3),
                    "gflops": round(full_batch / dt * TRAIN_GFLOP_PER_IMG, 1)}
            runs.append(full)
            del img64, tsd
        except (MemoryError, RuntimeError) as e:
            print(f"[bench] cpu baseline at batch {full_batch} not possible on this host: {e!r}", file=sys.stderr, flush=True)
    best = full or max((r for r in runs if r["mode"].startswith("train")), key=lambda r: r["img_s"])
    frac = "the bench workload's own batch" if best["batch"] == full_batch else f"the bench workload at 1/{max(full_batch // best['batch'], 1)} of its batch"
    return {"value": best["img_s"], "unit": "spectrogram-images/s", "cores": best["threads"], "kind": "port",
            "sample": f"yolo11s {imgsz}x{imgsz} at batch {best['batch']} ({frac}): train fwd+loss+bwd, "
                      f"{best['iters']} timed + 1 warm-up iterations, oracle/yolo11_ref.py + loss_ref.py, fp32, {best['threads']} torch threads",
            "host_cpus": ncpu, "runs": runs, "wall_s": round(time.perf_counter() - t_start, 1)}


def measured_peaks(dev):
    """The roofs as THIS GPU delivers them: a pure-MFMA f16 loop and a large device-to-device copy."""
    import ctypes as C

    from sy11 import _lib
    lib = _lib.load()
    wgs, iters = 256 * 8, 4000
    out = torch.empty(wgs * 4, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.sy11_peak_mfma_f16(wgs, 200, C.c_void_p(out.data_ptr()), st), "peak")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    _lib.check(lib.sy11_peak_mfma_f16(wgs, iters, C.c_void_p(out.data_ptr()), st), "peak")
    e1.record()
    torch.cuda.synchronize()
    tf = wgs * 4 * iters * 8 * 32768 / (e0.elapsed_time(e1) * 1e-3) / 1e12
    a = torch.empty(1 << 28, dtype=torch.float32, device=dev)       # 1 GiB
    b = torch.empty_like(a)
    b.copy_(a)
    e0.record()
    for _ in range(5):
        b.copy_(a)
    e1.record()
    torch.cuda.synchronize()
    gbs = 5 * 2 * a.numel() * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9
    return {"mfma_f16_tflops": round(tf, 1), "mfma_f16_vs_datasheet": round(tf / PEAK_TFLOPS["f16"], 3),
            "device_copy_GBps": round(gbs, 1), "device_copy_vs_datasheet": round(gbs / PEAK_HBM_GBS, 3),
            "how": "sy11_peak_mfma_f16 (2048 workgroups x 4 waves x 32000 v_mfma_f32_32x32x16_f16, registers only); torch copy_ of 1 GiB (read + write bytes)"}


def predict_val_leg(model, img, batch, nc, steps):
    """The predict / validation side at the bench's batch (BASELINE configs[0] scaled up; models/yolo/detect/val.py:93-106):
    fused eval forward -> Detect decode -> non_max_suppression(conf 0.001, iou 0.7, multi_label, max_det 300) on the HIP NMS.
    A random-init head scores every class ~1e-5 (Detect.bias_init), i.e. nothing would pass conf 0.001: the NMS is therefore fed
    SYNTHETIC decoded predictions with a stated candidate density (1 % of the anchor x class pairs above the threshold: 6 720
    candidates per image, a mid-training validation batch) next to the model's own decode timing."""
    from sy11 import _lib, ops as K
    from sy11.utils.ops import non_max_suppression
    dev = img.device
    with torch.no_grad():
        y, maps = model(img)
    nhwc = [m.permute(0, 2, 3, 1).contiguous() for m in maps]
    strides = [float(v) for v in model.stride]
    ev = lambda: torch.cuda.Event(enable_timing=True)          # noqa: E731
    K.detect_decode(nhwc, strides, nc)
    e0, e1 = ev(), ev()
    e0.record()
    for _ in range(10):
        K.detect_decode(nhwc, strides, nc)
    e1.record()
    torch.cuda.synchronize()
    dec_ms = e0.elapsed_time(e1) / 10
    dec_bytes = sum(m.numel() * 4 for m in nhwc) + y.numel() * 4
    g = torch.Generator(device=dev).manual_seed(7)
    A = y.shape[2]
    pred = torch.empty((batch, 4 + nc, A), device=dev)
    pred[:, 0:2] = 20 + 600 * torch.rand(batch, 2, A, generator=g, device=dev)
    pred[:, 2:4] = 8 + 120 * torch.rand(batch, 2, A, generator=g, device=dev) ** 2
    u = torch.rand(batch, nc, A, generator=g, device=dev)
    pred[:, 4:] = torch.where(u > 0.99, 0.001 + 0.9 * torch.rand(batch, nc, A, generator=g, device=dev) ** 3, 1e-4 * u)
    kw = dict(conf_thres=0.001, iou_thres=0.7, multi_label=True, max_det=300, nc=nc)
    out = non_max_suppression(pred.clone(), **kw)
    torch.cuda.synchronize()
    n_rep = max(min(steps // 4, 5), 2)
    _lib.PROFILE = []
    t0 = time.perf_counter()
    for _ in range(n_rep):
        out = non_max_suppression(pred.clone(), **kw)
    torch.cuda.synchronize()
    nms_ms = (time.perf_counter() - t0) / n_rep * 1e3
    prof, _lib.PROFILE = _lib.PROFILE, None
    k_ms = sum(e0.elapsed_time(e1) for name, e0, e1, _ in prof if name.startswith("sy11_nms_")) / n_rep
    cand = int((pred[:, 4:] > 0.001).sum().item()) / batch
    per_class = (pred[:, 4:] > 0.001).sum(2).double()             # candidates per (image, class): one bit matrix each
    pairs = float((per_class * per_class).sum().item()) / 2       # IoU evaluations of the bit-matrix kernels (upper triangles)
    return {"workload": "predict / val side at bs 64: Detect decode (model output) + non_max_suppression(conf 0.001, iou 0.7, multi_label, max_det 300) "
                        "on synthetic decoded predictions, 1 % of anchor x class pairs above conf",
            "candidates_per_image": round(cand, 1), "kept_per_image": round(sum(len(o) for o in out) / batch, 1),
            "detect_decode": {"kernel": "detect_decode_kernel", "ms": round(dec_ms, 4), "algorithmic_bytes": dec_bytes, "bound": "hbm",
                              "achieved": round(dec_bytes / dec_ms / 1e6, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                              "frac": round(dec_bytes / dec_ms / 1e6 / PEAK_HBM_GBS, 4)},
            "nms": {"kernel": "nms_candidates_kernel x2 (threshold + ordered compaction over the (B, 4 + nc, A) tensor) + nms_mask_seg_kernel + "
                              "nms_sweep_seg_kernel (one bit matrix per (image, class), all in one launch pair)",
                    "wrapper_ms_per_batch": round(nms_ms, 3), "kernel_ms_per_batch": round(k_ms, 3),
                    "images_per_s": round(batch / (nms_ms * 1e-3), 1), "iou_pairs_per_batch": int(pairs),
                    "note": "the kernels are 6 % of the wrapper now: the rest is one 64-bit key sort, gathers and three host reads (candidate "
                            "counts, longest segment, survivors per image); r03 tested every pair of an image's candidates (1.4 G pairs, 1.4 ms), "
                            "classes shifted by max_wh never intersect, so per-class matrices keep the same set bit for bit"}}


def family_of(name, meta):
    if name.startswith("sy11_conv2d") and meta and meta.get("groups", 1) > 1:
        return "depthwise conv"          # depthwise / grouped calls run the direct kernels: kept out of the dense families
    for fam, spec in FAMILIES.items():
        if name in spec["calls"]:
            return fam
    return name


def timed_steps(step, fence, steps, warmup):
    for _ in range(warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    fence()
    return time.perf_counter() - t0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU")
    ap.add_argument("--imgsz", type=int, default=640)
    ap.add_argument("--model", default="yolo11s.yaml")
    ap.add_argument("--nc", type=int, default=80, help="classes of the Detect head (2 for the fusion variant of configs[4])")
    ap.add_argument("--dtype", default="f16", choices=["f16", "f32"])
    ap.add_argument("--mode", default="train", choices=["train", "fwd"])
    ap.add_argument("--no-stft", action="store_true", help="feed resident (B,3,H,W) images instead of IQ")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-fwd-leg", action="store_true", help="skip the forward-only (configs[1]) leg reported next to the headline")
    ap.add_argument("--no-extras", action="store_true", help="skip the f32 train line and the configs[4] model line")
    ap.add_argument("--no-graphs", action="store_true", help="launch every kernel individually instead of hipGraph replay")
    ap.add_argument("--leg", default="", choices=["", "predict_val"], help="run ONE reporting leg only and print its JSON (for rocprofv3 profiles of that leg)")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python3 bench.py --gpus N` without a launcher: spawn one CHILD per GPU (engine/trainer.py:170-207, utils/dist.py:25-66).
        # Nothing in this parent has touched the GPU (importing torch does not), the children are fresh interpreters and never an
        # exec over a process that initialised the GPU; rank 0's JSON line is the children's stdout, passed through unchanged.
        from sy11.engine.ddp import launch
        codes = launch([str(Path(__file__).resolve()), *sys.argv[1:]], a.gpus)
        raise SystemExit(max(codes))

    # The contract is ONE JSON line on stdout.  Libraries write there too (gloo announces every group it connects — the trainer's
    # host-side control group included — "[Gloo] Rank 0 is connected to 1 peer ranks"): from here on file descriptor 1 IS stderr, and
    # the JSON line goes to the saved original.  (The self-launching parent above returns before this point: its children inherit the
    # real stdout.)
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    def emit(obj):
        os.write(json_fd, (json.dumps(obj) + "\n").encode())

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

    def make(model_yaml, nc, dtype, with_producer, ws=1):
        model = DetectionModel(model_yaml, nc=nc, verbose=False)
        producer = SpectrogramProducer(dev, n_frames=a.imgsz, n_mel=a.imgsz) if with_producer else None
        tr = DetectionTrainer(model, batch_size=a.batch, device=dev, overrides={"amp": dtype == "f16"}, world_size=ws,
                              producer=producer, graphs=not a.no_graphs)
        labels = synthetic_labels(a.batch, 100 + rank, dev, nc=nc)
        if producer is not None:
            data = {"iq": synthetic_iq(a.batch, producer.n_samples, 1 + rank, dev)}
        else:
            data = {"img": torch.rand(a.batch, 3, a.imgsz, a.imgsz, device=dev)}
        return tr, data, labels

    if a.leg == "predict_val":                              # the predict / val side alone: fused eval forward + decode + batched NMS
        from sy11.engine import enable_graphs
        m = DetectionModel(a.model, nc=a.nc, verbose=False).to(dev).eval()
        m._sy11_dtype = torch.float16 if a.dtype == "f16" else torch.float32
        m.fuse()
        enable_graphs(m)
        img = torch.rand(a.batch, 3, a.imgsz, a.imgsz, device=dev)
        with torch.no_grad():
            for _ in range(max(a.warmup, 3)):
                m(img)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(a.steps):
                m(img)
            torch.cuda.synchronize()
        fwd_ms = (time.perf_counter() - t0) / a.steps * 1e3
        leg = predict_val_leg(m, img, a.batch, a.nc, a.steps)
        leg["fused_eval_forward_ms"] = round(fwd_ms, 3)
        leg["val_images_per_s_forward_plus_nms"] = round(a.batch / ((fwd_ms + leg["nms"]["wrapper_ms_per_batch"]) * 1e-3), 1)
        emit(leg)
        return

    tr, data, labels = make(a.model, a.nc, a.dtype, not a.no_stft, world)

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

    # set-up, before the W warm-up steps and outside every timed region: the trainer runs the first two steps of an input signature
    # eagerly (the tile autotuner measures on the first) and captures its hipGraphs on the third — with a small --warmup the timed
    # steps would otherwise include the measuring and the capture
    for _ in range(3):
        step()
    dt = timed_steps(step, fence, a.steps, a.warmup)
    if rank == 0:
        print(f"[bench] headline: {a.steps} steps, {dt / a.steps * 1e3:.2f} ms/step", file=sys.stderr, flush=True)
    if world > 1:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = tmax.item()
    ms = dt / a.steps * 1e3
    value = a.batch * world * a.steps / dt
    exposed = None
    if world > 1:
        # the gradient exchange as the launch stream sees it: events around the hook's collective(s) (the stream-side wait for RCCL
        # included) over 5 more steps — what of the all-reduce is NOT hidden behind compute
        from sy11.engine import module_post_backward
        hk = module_post_backward.get(id(tr.model.__dict__.get("_sy11_grads")))
        if hk is not None:
            hk.timing = []
            for _ in range(5):
                step()
            torch.cuda.synchronize()
            spans = [e0.elapsed_time(e1) for e0, e1 in hk.timing]
            hk.timing = None
            per_step = sum(spans) / 5 if spans else 0.0
            t_all = torch.tensor([per_step], device=dev, dtype=torch.float64)
            dist.all_reduce(t_all, op=dist.ReduceOp.MAX)
            exposed = {"allreduce_ms_per_step_on_launch_stream": round(t_all.item(), 3), "bytes": int(tr.grad_store.flat.numel() * 4),
                       "overlap": bool(ddp.OVERLAP), "how": "HIP events on the launch stream around the collective calls of the gradient hook, max over ranks"}
    fwd_gflop = FUSION_FWD_GFLOP_PER_IMG if "fusion" in a.model else FWD_GFLOP_PER_IMG
    train_gflop = 3 * fwd_gflop - 0.177 if "fusion" in a.model else TRAIN_GFLOP_PER_IMG

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
                    "conv_tflops": round(a.batch / fdt * fwd_gflop / 1e3, 2),
                    "conv_roofline_frac": round(a.batch / fdt * fwd_gflop / 1e3 / PEAK_TFLOPS[a.dtype], 4)}

    roof = None
    if world > 1 and not a.no_roofline:
        step()                                                  # every rank takes the extra step: its gradient all-reduce must be matched
    if rank == 0 and not a.no_roofline:
        # one instrumented step: every C-ABI launch bracketed by events on the launch stream.  The launches are issued behind a
        # ~60 ms head-start delay on the same stream, so the GPU works through an already filled queue: an event pair then spans
        # the kernel(s) of the call, not the host's launch latency (r01: eager event timing read 9.8 ms for a family that takes
        # 8.8 ms under graph replay)
        from sy11.engine import module_post_backward
        import sy11.engine as _engine
        # ... and ONE stream: in the headline run the filter gradients are launched on a second stream and overlap the main chain
        # (every kernel of both streams then runs slower than alone, the step faster).  A kernel's roofline position is the
        # kernel's own: this instrumented step launches everything on the launch stream, as `profiles/<round>/z_serial_*`
        # (the same command with SY11_WGRAD_STREAM=0) does; `z_final_*` is the headline command as it runs.
        _side_was, _engine._SIDE_WGRAD = _engine._SIDE_WGRAD, False
        tr.model.__dict__.pop("_sy11_graph_cfg", None)          # per-launch events need individually launched kernels
        store = tr.model.__dict__.get("_sy11_grads")
        hook = module_post_backward.pop(id(store), None) if (store is not None and world > 1) else None   # the collective happened above
        # the two steps below run on rank 0 ALONE (the other ranks are past their last collective): no gradient hook (popped above) and
        # none of the trainer's per-step control collectives either (ddp.share_tuner_picks_if_any is a MAX all-reduce every step since r04)
        _dp_was, tr.data_parallel = tr.data_parallel, False
        step()                                                  # eager once (allocator warm, no capture bookkeeping in the timed step)
        torch.cuda.synchronize()
        _lib.PROFILE = []
        torch.cuda._sleep(int(0.06 * 2.0e9))
        empty = []
        for _ in range(64):                                     # what an event pair costs by itself on this stream (subtracted below)
            c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            c0.record()
            c1.record()
            empty.append((c0, c1))
        step()
        torch.cuda.synchronize()
        prof, _lib.PROFILE = _lib.PROFILE, None
        _engine._SIDE_WGRAD = _side_was
        pair_ms = sorted(c0.elapsed_time(c1) for c0, c1 in empty)[len(empty) // 2]
        if hook is not None:
            module_post_backward[id(store)] = hook
        tr.data_parallel = _dp_was
        fam = {}
        for name, e0, e1, meta in prof:
            f = fam.setdefault(family_of(name, meta), {"ms": 0.0, "n": 0, "flops": 0.0, "bytes": 0.0, "by_call": {}})
            t = max(e0.elapsed_time(e1) - 0.5 * pair_ms, 0.0)     # of the two event records that bracket a call, about one falls inside the interval
            f["ms"] += t
            f["n"] += 1
            c = f["by_call"].setdefault(name, {"ms": 0.0, "n": 0, "bytes": 0.0, "flops": 0.0})
            c["ms"] += t
            c["n"] += 1
            if meta and name.startswith(("sy11_conv2d", "sy11_stem", "sy11_bn_")):
                for d in (f, c):
                    d["flops"] += meta["flops"]
                    d["bytes"] += meta["bytes"]
        dump = os.environ.get("SY11_DUMP_LAUNCHES")
        if dump:
            with open(dump, "w") as fh:
                for name, e0, e1, meta in prof:
                    isconv = name.startswith(("sy11_conv2d", "sy11_stem"))
                    fh.write(json.dumps({"name": name, "ms": round(e0.elapsed_time(e1), 4),
                                         "gflop": round((meta or {}).get("flops", 0) / 1e9, 3) if isconv else 0,
                                         "mbytes": round((meta or {}).get("bytes", 0) / 1e6, 3) if isconv else 0,
                                         "desc": (meta or {}).get("desc") if isconv else None}) + "\n")
        name, f = max(fam.items(), key=lambda kv: kv[1]["ms"])
        peak = PEAK_TFLOPS[a.dtype]
        secs = f["ms"] * 1e-3
        ach_tf = f["flops"] / secs / 1e12 if secs > 0 else 0.0
        ach_gb = f["bytes"] / secs / 1e9 if secs > 0 else 0.0
        ai = f["flops"] / max(f["bytes"], 1.0)
        # which roof binds the family: its arithmetic intensity (algorithmic FLOPs / algorithmic bytes) against the ridge
        # peak_flops / peak_bandwidth.  yolo11s' conv families sit at ~150 FLOP/B in f16, below the 312 FLOP/B ridge -> HBM.
        hbm_bound = f["bytes"] > 0 and ai < peak * 1e12 / (PEAK_HBM_GBS * 1e9)
        traffic = None
        tj = PROFILE_DIR / "traffic.json"
        if tj.exists():                                          # HBM bytes of the same family from the rocprofv3 PMC passes
            tjd = json.loads(tj.read_text())
            ft = tjd.get("families", {}).get(name)
            if tjd.get("csrc_sha16") != kernel_source_sha16():   # measured on other kernel sources: not this build's traffic
                print(f"[bench] {tj} was measured on csrc {tjd.get('csrc_sha16')}, this tree is {kernel_source_sha16()}: traffic = null", file=sys.stderr)
                ft = None
            if ft:
                traffic = {"bytes_per_step": round(ft["GB_per_step"] * 1e9), "kernel_launches_per_step": ft["launches_per_step"],
                           "bytes_per_kernel_launch": round(ft["GB_per_step"] * 1e9 / max(ft["launches_per_step"], 1)),
                           "vs_algorithmic": round(ft["GB_per_step"] * 1e9 / max(f["bytes"], 1.0), 3),
                           "source": f"profiles/{PROFILE_DIR.name}/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, gfx950 corrections)",
                           "commit": json.loads(tj.read_text()).get("commit"), "commands": json.loads(tj.read_text()).get("commands")}
        dominant = None
        kcsv = PROFILE_DIR / "z_serial_per_step_kernels.csv"
        if not kcsv.exists():
            kcsv = PROFILE_DIR / "z_final_per_step_kernels.csv"
        if kcsv.exists():                                        # hottest kernel SYMBOL of the family in the committed graph-replay trace
            import csv
            rows = [r for r in csv.DictReader(open(kcsv)) if r["family"] == name]
            if rows:
                top = max(rows, key=lambda r: float(r["ms_per_step"]))
                dominant = {"symbol": top["kernel"], "launches_per_step": float(top["launches_per_step"]), "ms_per_step": float(top["ms_per_step"]),
                            "avg_us_per_launch": float(top["avg_us_per_launch"]), "source": f"profiles/{PROFILE_DIR.name}/{kcsv.name} (rocprofv3 --kernel-trace --stats of this command, kernels one at a time: SY11_WGRAD_STREAM=0)"}
        roof = {"kernel": name, "kernel_symbols": list(FAMILIES.get(name, {}).get("symbols", ())), "dominant_symbol": dominant,
                "bound": "hbm" if hbm_bound else "mfma",
                "achieved": round(ach_gb, 1) if hbm_bound else round(ach_tf, 2), "peak": PEAK_HBM_GBS if hbm_bound else peak,
                "unit": "GB/s" if hbm_bound else "TFLOP/s",
                "frac": round((ach_gb / PEAK_HBM_GBS) if hbm_bound else (ach_tf / peak), 4),
                "traffic": traffic["bytes_per_step"] if traffic else None, "traffic_detail": traffic,
                "per": "training step (all launches of the family in one step; `achieved` = algorithmic bytes of those launches / their summed duration)",
                "arithmetic_intensity_flop_per_byte": round(ai, 1), "mfma_tflops": round(ach_tf, 2), "mfma_frac": round(ach_tf / peak, 4),
                "family_ms_per_step": round(f["ms"], 3), "c_abi_calls_per_step": f["n"],
                "algorithmic_bytes_per_step": round(f["bytes"]), "algorithmic_gflop_per_step": round(f["flops"] / 1e9, 1),
                "avg_call_ms": round(f["ms"] / max(f["n"], 1), 4),
                "by_call": {k: {"ms": round(v["ms"], 3), "calls": v["n"], "algorithmic_GB": round(v["bytes"] / 1e9, 3),
                                "GBps": round(v["bytes"] / max(v["ms"], 1e-9) / 1e6, 1), "tflops": round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 1)}
                            for k, v in f["by_call"].items()},
                "families_ms": {k: round(v["ms"], 3) for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["ms"])},
                "timing": "HIP events around each C-ABI call of one eager step issued behind a 60 ms head-start delay on the launch stream, "
                          f"minus half the span of an empty event pair ({pair_ms * 1e3:.1f} us); every kernel on ONE stream for this "
                          "step (the headline run overlaps the filter gradients on a second stream: each kernel slower, the step faster)"}

    peaks = measured_peaks(dev) if (rank == 0 and not a.no_roofline) else None

    if fwd_only is not None:
        # the deployed forward: BatchNorm folded into the convs (model.fuse()), eval mode, Detect decode included.  Last thing
        # this process does with the headline model (fusing is destructive); reported inside `forward_only`, never part of `value`.
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
                                  "ms_per_step": round(pdt * 1e3, 3), "conv_tflops": round(a.batch / pdt * fwd_gflop / 1e3, 2),
                                  "conv_roofline_frac": round(a.batch / pdt * fwd_gflop / 1e3 / PEAK_TFLOPS[a.dtype], 4)}
        try:
            fwd_only["predict_val"] = predict_val_leg(m, img, a.batch, a.nc, a.steps)
            pv = fwd_only["predict_val"]
            pv["val_images_per_s_forward_plus_nms"] = round(a.batch / (pdt + pv["nms"]["wrapper_ms_per_batch"] * 1e-3), 1)
        except Exception as e:                                   # a reporting leg must never take the headline down
            fwd_only["predict_val"] = {"error": repr(e)}

    extra = None
    if rank == 0 and world == 1 and a.mode == "train" and not a.no_extras and a.model == "yolo11s.yaml":
        del tr
        torch.cuda.empty_cache()
        extra = {}
        for key, (yaml_, nc, dtype, what, gf) in {
                "f32_train": ("yolo11s.yaml", 80, "f32", "the headline workload with exact-f32 MFMA and no AMP (the dtype of the 1e-3 parity bar)", TRAIN_GFLOP_PER_IMG),
                "configs4_model": ("yolo11s_fusion_sand3_new.yaml", 2, "f16", "configs[4]'s model (Spectrogram-YOLOv11, 6 824 734 parameters, nc = 2), f16 train step on one GPU",
                                   3 * FUSION_FWD_GFLOP_PER_IMG - 0.177)}.items():
            t2_, d2, l2 = make(yaml_, nc, dtype, True)
            k = max(a.steps // 3, 5)
            edt = timed_steps(lambda: t2_.train_step({**d2, **l2}), torch.cuda.synchronize, k, 4)
            print(f"[bench] extra {key}: {edt / k * 1e3:.2f} ms/step", file=sys.stderr, flush=True)
            extra[key] = {"workload": what, "value": round(a.batch * k / edt, 1), "unit": "spectrogram-images/s", "ms_per_step": round(edt / k * 1e3, 3),
                          "steps": k, "dtype": dtype, "conv_tflops": round(a.batch * k / edt * gf / 1e3, 2)}
            del t2_, d2, l2
            torch.cuda.empty_cache()

    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        cpu = cpu_baseline()

    if rank == 0:
        gflop = train_gflop if a.mode == "train" else fwd_gflop
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
                       "parallelism": f"dp{world}", "weights": "random-init", "nc": a.nc},
            "conv_tflops": round(value * gflop / 1e3, 2),
            "conv_roofline_frac": round(value * gflop / 1e3 / PEAK_TFLOPS[a.dtype], 4),
            "roofline": roof, "peaks": peaks, "cpu_baseline": cpu, "forward_only": fwd_only, "extra": extra, "gradient_exchange": exposed,
        }
        emit(out)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
