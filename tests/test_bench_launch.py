"""bench.py's multi-rank launch contract, as far as a box without a GPU can show it: `--gpus N` without WORLD_SIZE
starts N ranks itself (torch.distributed.run, children -- the parent never initialises the GPU), a failing rank makes
the launch exit non-zero, and a mismatch between --gpus and the launcher's WORLD_SIZE is refused."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra, timeout=300):
    env = dict(os.environ, **env_extra)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        if k not in env_extra:
            env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                          timeout=timeout)


def test_gpus_flag_spawns_ranks_and_propagates_failure():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("the no-GPU failure path")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--train-steps", "0"],
             {"MAL_BENCH_BACKEND": "gloo"})
    assert r.returncode != 0
    # both children got as far as bench.py's own device check: they were really started as ranks
    assert r.stderr.count("bench.py needs a HIP device") >= 2, r.stderr[-2000:]


def test_world_size_mismatch_is_refused():
    r = _run(["--gpus", "1"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_nccl_launch_needs_enough_devices():
    r = _run(["--gpus", "2"], {})
    assert r.returncode != 0 and "GPU(s) visible" in r.stderr
