"""bench.py's launcher logic on CPU: `--gpus N` without a launcher re-executes under torch.distributed.run with N ranks
(`--gpus N` used to be ignored), and asks for more GPUs than the box has fail loudly."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_single_gpu_needs_no_launcher(monkeypatch):
    monkeypatch.delenv("RANK", raising=False)
    assert bench.respawn_command(bench.parse(["--gpus", "1"]), ["--gpus", "1"], 8) is None


def test_gpus_n_spawns_n_ranks(monkeypatch):
    monkeypatch.delenv("RANK", raising=False)
    argv = ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    cmd = bench.respawn_command(bench.parse(argv), argv, 8)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-len(argv):] == argv and cmd[-len(argv) - 1].endswith("bench.py")


def test_a_rank_does_not_respawn(monkeypatch):
    monkeypatch.setenv("RANK", "2")
    assert bench.respawn_command(bench.parse(["--gpus", "4"]), ["--gpus", "4"], 8) is None


def test_too_few_devices_fails_loudly(monkeypatch):
    monkeypatch.delenv("RANK", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.respawn_command(bench.parse(["--gpus", "8"]), ["--gpus", "8"], 1)
    assert "only 1 gfx950" in str(e.value)
