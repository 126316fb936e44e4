"""GPU (needs >= 2 visible GPUs — skips on the one-GPU test box): the data-parallel path over RCCL, one process per GPU, as
`bench.py --gpus N` and the trainer run it (the two-rank test), and — on the one-GPU box — the one-rank rehearsal of the same
collectives through a world-size-1 RCCL communicator against the single-process trainer.  Both flavours of the gradient exchange: the default (ONE all-reduce of the flat
gradient buffer after the backward graph) and SY11_DDP_OVERLAP=1 (two-bucket exchange: the first bucket is reduced on RCCL's
stream beside the second backward graph).  After 8 steps on different batches the ranks must hold bit-identical weights, and the
overlapped flavour must reproduce the default one bit for bit (the same sums in the same order, only scheduled differently)."""
import os
import sys
from pathlib import Path

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]

SCRIPT = """
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.path.join(r"{root}", "spectrogram-yolov11_amd")); sys.path.insert(0, r"{root}")
from sy11 import _lib
from sy11.engine import ddp
from sy11.engine.trainer import DetectionTrainer
from sy11.nn.tasks import DetectionModel
rank, local, world = ddp.setup_process_group("nccl")
dev = torch.device("cuda", local)
torch.manual_seed(5 + rank)
m = DetectionModel("yolo11n.yaml", nc=80, verbose=False)
tr = DetectionTrainer(m, batch_size=8, device=dev, overrides={{"amp": True, "nbs": 8 * world, "warmup_epochs": 0}}, world_size=world, graphs=True)
losses = []
w0 = tr.flat.flat.clone()
for i in range(8):
    g = torch.Generator().manual_seed(100 * rank + i)
    b = {{"img": torch.rand(8, 3, 256, 256, generator=g).to(dev), "batch_idx": torch.arange(8.0).to(dev),
         "cls": torch.randint(0, 80, (8, 1), generator=g).float().to(dev), "bboxes": (0.3 + 0.3 * torch.rand(8, 4, generator=g)).to(dev)}}
    losses.append(float(tr.train_step(b)[0]))
torch.cuda.synchronize()
e = next(iter(tr.model.__dict__["_sy11_graph_cfg"]["entries"].values()))
assert (e.g_bwd2 is not None) == (os.environ.get("SY11_DDP_OVERLAP", "0") == "1")
mine = tr.flat.flat.clone()
theirs = mine.clone(); dist.broadcast(theirs, 0)
assert torch.equal(mine, theirs), float((mine - theirs).abs().max())
moved = float((mine != w0).float().mean())             # a GradScaler that skipped every step would make the comparison above vacuous
assert moved > 0.5, f"only {{moved:.3f}} of the weights changed in eight steps"
picks = _lib.tune_export()
box = [picks]; dist.broadcast_object_list(box, 0)
assert sorted(picks[i:i + 16] for i in range(0, len(picks), 16)) == sorted(box[0][i:i + 16] for i in range(0, len(box[0]), 16))
if rank == 0:
    torch.save({{"flat": mine.cpu(), "losses": losses}}, r"{out}")
dist.barrier(); dist.destroy_process_group()
"""


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one GPU per rank: fewer than 2 GPUs visible")
def test_two_ranks_over_rccl_identical_weights_both_exchange_flavours(tmp_path):
    sys.path.insert(0, str(ROOT / "spectrogram-yolov11_amd"))
    from sy11.engine import ddp
    outs = []
    for overlap in ("0", "1"):
        out = tmp_path / f"w{overlap}.pt"
        script = tmp_path / f"ddp_nccl_{overlap}.py"
        script.write_text(SCRIPT.format(root=str(ROOT), out=str(out)))
        # rank 0 measures tile configurations in the first flavour; the second replays those picks so that both run the same kernels
        env = dict(os.environ, OMP_NUM_THREADS="2", SY11_DDP_OVERLAP=overlap, SY11_TUNE="0")
        assert ddp.launch([str(script)], 2, env=env, timeout=900) == [0, 0]
        outs.append(torch.load(out))
    # f32 atomics reorder sums run to run: the two flavours agree to rounding, not bit for bit
    a, b = outs[0]["flat"], outs[1]["flat"]
    assert (a - b).norm().item() <= 2e-3 * a.norm().item()
    assert all(abs(x - y) <= 2e-2 * abs(x) for x, y in zip(outs[0]["losses"], outs[1]["losses"]))


REHEARSAL = """
import os, sys, torch
sys.path.insert(0, os.path.join(r"{root}", "spectrogram-yolov11_amd")); sys.path.insert(0, r"{root}")
import torch.distributed as dist
from sy11.engine import ddp
from sy11.engine.trainer import DetectionTrainer
from sy11.nn.tasks import DetectionModel
rank, local, world = ddp.setup_process_group("nccl" if ddp.REHEARSE else None)
assert world == 1 and dist.is_initialized() == ddp.REHEARSE
if ddp.REHEARSE:
    assert dist.get_backend() == "nccl" and ddp.active()
dev = torch.device("cuda", 0)
torch.manual_seed(5)
m = DetectionModel("yolo11n.yaml", nc=80, verbose=False)
tr = DetectionTrainer(m, batch_size=8, device=dev, overrides={{"amp": True, "nbs": 8, "warmup_epochs": 0}}, world_size=1, graphs=True)
assert tr.data_parallel == ddp.REHEARSE
calls = []
if ddp.REHEARSE:                                         # count what really reaches the backend
    real = dist.all_reduce
    def counted(t, *a, **k):
        if t.is_cuda:
            calls.append(t.numel())
        return real(t, *a, **k)
    dist.all_reduce = counted
losses = []
for i in range(8):
    g = torch.Generator().manual_seed(100 + i)
    b = {{"img": torch.rand(8, 3, 256, 256, generator=g).to(dev), "batch_idx": torch.arange(8.0).to(dev),
         "cls": torch.randint(0, 80, (8, 1), generator=g).float().to(dev), "bboxes": (0.3 + 0.3 * torch.rand(8, 4, generator=g)).to(dev)}}
    losses.append(float(tr.train_step(b)[0]))
torch.cuda.synchronize()
e = next(iter(tr.model.__dict__["_sy11_graph_cfg"]["entries"].values()))
assert (e.g_bwd2 is not None) == (ddp.REHEARSE and os.environ.get("SY11_DDP_OVERLAP", "0") == "1")
torch.save({{"flat": tr.flat.flat.cpu(), "ema": tr.ema.ema_state.flat.cpu(), "losses": losses, "calls": calls}}, r"{out}")
if ddp.REHEARSE:
    dist.barrier(); dist.destroy_process_group()
"""


def test_one_rank_rehearsal_over_rccl_reproduces_the_single_process_trainer(tmp_path):
    """The one-GPU box's RCCL run: SY11_DDP_REHEARSE=1 sends every collective of the data-parallel trainer (parameter broadcast,
    control-group flag, tuner-pick broadcast, the flat-gradient all-reduce after the backward graph — and with SY11_DDP_OVERLAP=1
    the two-bucket exchange on RCCL's own stream beside the second backward graph) through a world-size-1 RCCL communicator.
    A sum over one rank is the identity: eight graph-replayed AMP steps must leave the weights, the EMA and the losses BIT-identical
    to the same trainer without a process group (ordered reductions, tuner off), and the gradient all-reduces must really have
    been issued on device tensors (>= one per step)."""
    import subprocess
    from sy11.engine import ddp
    outs = {}
    for name, rehearse, overlap in (("plain", "0", "0"), ("rccl", "1", "0"), ("rccl_overlap", "1", "1")):
        out = tmp_path / f"{name}.pt"
        script = tmp_path / f"{name}.py"
        script.write_text(REHEARSAL.format(root=str(ROOT), out=str(out)))
        env = dict(os.environ, OMP_NUM_THREADS="2", SY11_TUNE="0", SY11_DETERMINISTIC="1", SY11_DDP_REHEARSE=rehearse, SY11_DDP_OVERLAP=overlap,
                   WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(ddp.free_port()),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, f"{name}: {r.stderr[-2000:]}"
        outs[name] = torch.load(out)
    ref = outs["plain"]
    assert ref["calls"] == []
    for name in ("rccl", "rccl_overlap"):
        got = outs[name]
        assert torch.equal(got["flat"], ref["flat"]), f"{name}: weights differ by {(got['flat'] - ref['flat']).abs().max().item():.3e}"
        assert torch.equal(got["ema"], ref["ema"]), f"{name}: EMA differs"
        assert got["losses"] == ref["losses"], f"{name}: losses {got['losses']} vs {ref['losses']}"
        assert len(got["calls"]) >= 8 and max(got["calls"]) > 1_000_000, f"{name}: gradient all-reduces seen by the backend: {got['calls'][:12]}"
    assert len(outs["rccl_overlap"]["calls"]) > len(outs["rccl"]["calls"])          # the bucketed flavour issues several calls per step
