"""CPU: bench.py's launcher logic — `python bench.py --gpus N` (N > 1) without WORLD_SIZE must become the reference's launch line
(README.md:195 `torchrun --nproc_per_node=8 ...`), one rank per GPU, rendezvous on 127.0.0.1."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_main(monkeypatch, argv, env_world=None):
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    def fake_execv(exe, cmd):
        seen["exe"], seen["cmd"] = exe, list(cmd)
        raise SystemExit(0)
    monkeypatch.setattr(os, "execv", fake_execv)
    monkeypatch.setattr(sys, "argv", ["bench.py"] + argv)
    if env_world is None:
        monkeypatch.delenv("WORLD_SIZE", raising=False)
    else:
        monkeypatch.setenv("WORLD_SIZE", str(env_world))
    try:
        bench.main()
    except SystemExit:
        pass
    except AssertionError as e:       # no GPU here: main() stops at its "needs MI355X GPUs" assert when it did not respawn
        seen["assert"] = str(e)
    return seen


def test_gpus_n_without_a_launcher_respawns_under_torch_distributed_run(monkeypatch):
    seen = _run_main(monkeypatch, ["--gpus", "8", "--steps", "3", "--warmup", "1"])
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "3", "--warmup", "1"]
    assert os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"


@pytest.mark.parametrize("argv,world", [(["--gpus", "1"], None), (["--gpus", "8"], 8)])
def test_no_respawn_for_one_gpu_or_under_a_launcher(monkeypatch, argv, world):
    seen = _run_main(monkeypatch, argv, env_world=world)
    assert "cmd" not in seen
