"""bench.py's launch logic (no GPU): --gpus N without a launcher spawns torch.distributed.run as a child process before torch
is imported; --gpus that disagrees with WORLD_SIZE is refused instead of silently recording a 1-rank run as N GPUs."""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_self_launch_builds_the_drivers_command(monkeypatch):
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return subprocess.CompletedProcess(cmd, 7)

    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(subprocess, "run", fake_run)
    rc = bench.maybe_self_launch(argparse.Namespace(gpus=4), ["--gpus", "4", "--steps", "3"])
    assert rc == 7  # the child's exit code is passed through
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-5:] == [os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "3"]
    assert seen["env"]["MASTER_ADDR"] == "127.0.0.1"


def test_single_gpu_and_launched_ranks_do_not_relaunch(monkeypatch):
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert bench.maybe_self_launch(argparse.Namespace(gpus=1), []) is None
    monkeypatch.setenv("WORLD_SIZE", "8")
    assert bench.maybe_self_launch(argparse.Namespace(gpus=8), []) is None


def test_gpus_must_agree_with_world_size(monkeypatch, capsys):
    monkeypatch.setenv("WORLD_SIZE", "1")
    assert bench.maybe_self_launch(argparse.Namespace(gpus=8), []) == 2
    assert "WORLD_SIZE=1" in capsys.readouterr().err


def test_bench_does_not_touch_torch_before_the_launch_decision():
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src[: src.index("rc = maybe_self_launch")]
    assert "import torch" not in head.split("def main")[1]
