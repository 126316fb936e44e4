"""`python3 bench.py --gpus N` without a launcher spawns its own ranks (the reference spawns them itself:
ultralytics/engine/trainer.py:170-207, utils/dist.py:25-66).  CPU-side: the argument path only — what is handed to
``sy11.engine.ddp.launch`` and that the parent never reaches the GPU code."""
import importlib
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


def _bench():
    sys.path.insert(0, str(ROOT))
    return importlib.import_module("bench")


def test_gpus_n_without_world_size_spawns_children(monkeypatch):
    bench = _bench()
    from sy11.engine import ddp
    seen = {}

    def fake_launch(script_args, nproc, env=None, timeout=None):
        seen["args"], seen["n"] = list(script_args), nproc
        return [0] * nproc

    monkeypatch.setattr(ddp, "launch", fake_launch)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7", "--warmup", "2"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0
    assert seen["n"] == 4
    assert Path(seen["args"][0]).name == "bench.py" and seen["args"][1:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]


def test_launch_sets_rank_environment_and_reports_failures(tmp_path):
    """The real launcher on a trivial script: every child sees RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*; a failing rank raises."""
    from sy11.engine import ddp
    script = tmp_path / "child.py"
    script.write_text("import os, sys\n"
                      "open(sys.argv[1] + os.environ['RANK'], 'w').write(' '.join(os.environ[k] for k in "
                      "('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR')) + ' ' + str(int(os.environ['MASTER_PORT']) > 0))\n"
                      "sys.exit(int(sys.argv[2]) if os.environ['RANK'] == '1' else 0)\n")
    assert ddp.launch([str(script), str(tmp_path / "r"), "0"], 2, timeout=60) == [0, 0]
    assert (tmp_path / "r0").read_text() == "0 0 2 127.0.0.1 True" and (tmp_path / "r1").read_text() == "1 1 2 127.0.0.1 True"
    with pytest.raises(RuntimeError):
        ddp.launch([str(script), str(tmp_path / "q"), "3"], 2, timeout=60)
