"""GPU: the data-parallel path with REAL kernels — two ranks share cuda:0 over the gloo backend (one-GPU box; RCCL needs one
GPU per rank).  Each rank runs DetectionTrainer with hipGraph replay: the backward is captured as TWO graphs around the bucket
mark (SY11_DDP_OVERLAP=1; the default is one graph and ONE all-reduce after it), the first bucket's all-reduce is issued between
them, rank 0's tuner picks reach rank 1, and both ranks hold bit-identical weights after 8 steps (and the weights have moved) although they saw different
batches.  The RCCL flavour of this test (one GPU per rank) is tests/test_ddp_nccl_gpu.py; it skips on a one-GPU box."""
import os
import sys
import textwrap
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]

SCRIPT = textwrap.dedent("""
    import os, sys, torch, torch.distributed as dist
    sys.path.insert(0, os.path.join(r"{root}", "spectrogram-yolov11_amd")); sys.path.insert(0, r"{root}")
    from oracle import yolo11_ref as R
    from sy11 import _lib
    from sy11.engine import ddp
    from sy11.engine.trainer import DetectionTrainer
    from sy11.nn.tasks import DetectionModel
    rank, local, world = ddp.setup_process_group("gloo")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    torch.manual_seed(5 + rank)
    m = DetectionModel("yolo11n.yaml", nc=80, verbose=False)
    tr = DetectionTrainer(m, batch_size=4, device=dev, overrides={{"amp": True, "nbs": 8, "warmup_epochs": 0}}, world_size=world, graphs=True)
    assert tr.accumulate == 1 and (_lib.get_option("tune") == (1 if rank == 0 else 0))
    losses = []
    w0 = tr.flat.flat.clone()
    for i in range(8):
        g = torch.Generator().manual_seed(100 * rank + i)
        b = {{"img": torch.rand(4, 3, 128, 128, generator=g).to(dev), "batch_idx": torch.tensor([0., 1., 2., 3.]).to(dev),
             "cls": torch.randint(0, 80, (4, 1), generator=g).float().to(dev), "bboxes": (0.3 + 0.3 * torch.rand(4, 4, generator=g)).to(dev)}}
        losses.append(float(tr.train_step(b)[0]))
    entries = tr.model.__dict__["_sy11_graph_cfg"]["entries"]
    assert len(entries) == 1
    e = next(iter(entries.values()))
    # SY11_DDP_OVERLAP=1: the backward was captured in two parts around the bucket mark; default: one graph, one all-reduce after it
    assert (e.g_bwd2 is not None) == (os.environ.get("SY11_DDP_OVERLAP", "0") == "1")
    mine = tr.flat.flat.clone()
    theirs = mine.clone(); dist.broadcast(theirs, 0)
    assert torch.equal(mine, theirs), (mine - theirs).abs().max()
    moved = float((mine != w0).float().mean())         # a GradScaler that skipped every step would make the comparison above vacuous
    assert moved > 0.5, f"only {{moved:.3f}} of the weights changed in eight steps"

    picks = _lib.tune_export()
    box = [picks]; dist.broadcast_object_list(box, 0)
    assert sorted(picks[i:i + 16] for i in range(0, len(picks), 16)) == sorted(box[0][i:i + 16] for i in range(0, len(box[0]), 16)) and len(picks) > 0
    assert all(l == l for l in losses)
    dist.barrier(); dist.destroy_process_group()
    print("rank", rank, "ok", losses[-1])
""")


@pytest.mark.parametrize("overlap", ["0", "1"], ids=["one_allreduce_default", "two_bucket_overlap"])
def test_two_ranks_identical_weights(tmp_path, overlap):
    script = tmp_path / "ddp_gpu.py"
    script.write_text(SCRIPT.format(root=str(ROOT)))
    sys.path.insert(0, str(ROOT / "spectrogram-yolov11_amd"))
    from sy11.engine import ddp
    env = dict(os.environ, OMP_NUM_THREADS="2", SY11_TUNE="1", SY11_DDP_OVERLAP=overlap)
    assert ddp.launch([str(script)], 2, env=env, timeout=600) == [0, 0]


def test_bench_two_ranks_prints_exactly_one_json_line(tmp_path):
    """`python3 bench.py --gpus 2` as the driver runs it, rehearsed on one GPU (both ranks on cuda:0 over gloo): the self-launching parent
    returns 0, stdout carries EXACTLY one line and it is the JSON line of rank 0 (gloo announces every group it connects on stdout —
    the trainer's control group included — so bench.py writes its line to the saved descriptor and points fd 1 at stderr), and rank 0's
    own roofline leg, which runs after the other rank has left, issues no collective (r04: the per-step control all-reduce did)."""
    import json
    import subprocess
    env = dict(os.environ, SY11_DDP_BACKEND="gloo", SY11_FORCE_DEVICE="0", OMP_NUM_THREADS="2")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2", "--batch", "4", "--imgsz", "256",
                        "--no-stft", "--no-cpu-baseline", "--no-extras", "--no-fwd-leg"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.split("\n") if ln.strip()]
    assert len(lines) == 1, r.stdout[:2000]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0 and out["config"]["global_batch"] == 8
    assert out["gradient_exchange"] is not None and out["roofline"] is not None


def test_front_door_train_with_a_device_list_two_ranks(tmp_path):
    """`YOLO(cfg).train(data=yaml, device=[0, 1])` outside a launcher (engine/model.py:754-840 -> engine/trainer.py:170-207): the parent
    spawns one child per device before it has touched the GPU, the children shard the dataset (DistributedSampler semantics), sum their
    gradients every step, rank 0 validates and saves, the early-stop flag is broadcast, and the parent continues with best.pt and the
    per-epoch records.  Rehearsed with both ranks on cuda:0 over gloo (SY11_FORCE_DEVICE / SY11_DDP_BACKEND, read by
    ddp.setup_process_group); run in a fresh interpreter because the spawning parent must not have initialised the GPU."""
    import json
    import subprocess
    script = tmp_path / "front_door.py"
    script.write_text(textwrap.dedent(f"""
        import sys, json
        from pathlib import Path
        sys.path.insert(0, r"{ROOT}"); sys.path.insert(0, r"{ROOT / 'spectrogram-yolov11_amd'}")
        from tests.test_engine_flow_gpu import _dataset
        from sy11 import YOLO
        if __name__ == "__main__":
            root = Path(r"{tmp_path}")
            _dataset(root / "ds" / "train", 16, 96, 5)
            _dataset(root / "ds" / "val", 6, 96, 6)
            (root / "ds" / "data.yaml").write_text("path: .\\ntrain: train/images\\nval: val/images\\nnames:\\n  0: bright\\n  1: dark\\n")
            y = YOLO("yolo11n.yaml")
            hist = y.train(data=str(root / "ds" / "data.yaml"), epochs=2, batch=4, imgsz=96, workers=2, save_dir=str(root / "run"), close_mosaic=1,
                           warmup_epochs=0.5, device=[0, 1])
            print(json.dumps({{"epochs": len(hist), "names": y.model.names, "has_ckpt": y.ckpt is not None,
                              "files": sorted(p.name for p in (root / "run").iterdir())}}, default=str))
        """))
    env = dict(os.environ, SY11_DDP_BACKEND="gloo", SY11_FORCE_DEVICE="0", OMP_NUM_THREADS="2")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.split("\n") if ln.startswith("{")][-1])
    assert out["epochs"] == 2 and out["has_ckpt"] and {"last.pt", "results.json"} <= set(out["files"])
