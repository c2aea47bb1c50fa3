"""bench.py's launcher logic without a GPU: `--gpus N` starts its own ranks, picks the job BASELINE.json names for
several GPUs, and refuses to print a figure for a GPU count it did not run on."""
import os
import subprocess
import sys

from conftest import ROOT

if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def test_workload_resolution():
    import bench
    assert bench.resolve_workload(None, 1) == ("c3", 1_000_000, "weak")            # the N=1 headline: configs[2]
    assert bench.resolve_workload(None, 8) == ("c4", 1_250_000, "strong")          # configs[3]: 10M cells over 8 GPUs
    assert bench.resolve_workload(None, 2) == ("c4", 5_000_000, "strong")
    assert bench.resolve_workload("c5", 4) == ("c5", 2_500_000, "strong")          # configs[4]
    assert bench.resolve_workload("c3", 4) == ("c3", 1_000_000, "weak")
    assert bench.resolve_workload("c5", 1) == ("c5", 1_250_000, "weak")            # one GPU: the 8-GPU shard


def test_self_launch_command():
    import bench
    cmd = bench.self_launch(["--gpus", "4", "--steps", "3"], 4)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    i = cmd.index("--master-addr")
    assert cmd[i + 1] == "127.0.0.1" and 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    assert cmd[-4:] == ["--gpus", "4", "--steps", "3"] and cmd[-5].endswith("bench.py")


def _run(args, env_extra):
    env = dict(os.environ, **env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, env=env, timeout=600)


def test_mismatched_world_size_is_an_error():
    """Launched with WORLD_SIZE=1 but --gpus 2 (how the round-2 line came to say n_gpus: 1): non-zero exit, no JSON line."""
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and r.stdout.strip() == "" and "refusing" in r.stderr


def test_missing_gpus_are_an_error():
    """No (or too few) devices: exit code 3 instead of a line for fewer GPUs.  (Runs on the CPU-only build container.)"""
    import torch
    if torch.cuda.device_count() >= 1:
        import pytest
        pytest.skip("a GPU is visible")
    r = _run(["--gpus", "1", "--steps", "1", "--warmup", "0"], {})
    assert r.returncode == 3 and r.stdout.strip() == "" and "visible" in r.stderr
