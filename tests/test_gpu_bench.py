"""The driver's own command, run as a child process on the GPU box: `python bench.py --gpus 1 --steps K --warmup W` with
NO other flags must exit 0 and print ONE JSON line that carries the contract fields, the roofline block (HBM fraction
and the vector-ALU bound) and the CPU baseline.  Round 2 lost its headline to a bench.py edit that was never run with
the default arguments; this test is that run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, timeout=900):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True, text=True,
                       timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, "bench.py %s exited %d\n%s" % (" ".join(args), r.returncode, r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert lines, r.stderr[-2000:]
    return json.loads(lines[-1])


def _contract(d, steps, warmup):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == steps and d["warmup"] == warmup
    assert d["dtype"] == "f32" and d["unit"] == "images/s" and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["value"] > 0 and d["ms_per_step"] > 0
    assert isinstance(d["config"].get("workload"), str) and "model" not in d["config"]


def test_driver_command_default_flags():
    d = _bench(["--gpus", "1", "--steps", "2", "--warmup", "1"])
    _contract(d, 2, 1)
    assert "--temporal --distil" in d["config"]["workload"] and d["config"]["width"] == 640 and d["config"]["global_batch"] == 12
    assert abs(d["value"] - 12 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "valu") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert 0.05 < r["frac"] < 1.0 and r["kernel_ms"] > 0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert abs(r["achieved"] - 96 * 12 * 192 * 640 / (r["kernel_ms"] * 1e-3) / 1e9) <= 1e-6 * r["achieved"]
    # measured in THIS run from a graph holding only 64 back-to-back launches of the kernel (no committed file is read), in the
    # COLD regime: launch i works on batch i % 8, so no launch finds its operands in the Infinity Cache; the warm figure
    # (one batch replayed: what rounds 1-4 reported) sits beside it
    assert "64 back-to-back launches" in r["kernel_ms_how"] and r["launches_timed"] == 640, r["kernel_ms_how"]
    assert r["regime"] == "cold" and r["batches_rotated"] == 8 and "batch i % 8" in r["kernel_ms_how"]
    assert r["cold_kernel_ms"] == r["kernel_ms"] and abs(r["cold_frac"] - r["frac"]) < 1e-12
    assert r["warm_kernel_ms"] > 0 and 0.5 < r["warm_kernel_ms"] / r["cold_kernel_ms"] < 1.1 and 0.05 < r["warm_frac"] < 1.0
    assert r["eager_kernel_ms"] > 0 and 0.5 < r["kernel_ms"] / r["eager_kernel_ms"] < 1.2 and "replayed_rocprof" not in r
    # ... and so is the step: the timed steps rotate over 6 batches (value = the cold regime's), the warm replay beside it
    assert d["batches_rotated"] == 6 and d["config"]["regime"].startswith("cold")
    assert d["cold_ms_per_step"] == d["ms_per_step"] and d["cold_value"] == d["value"]
    assert d["warm_ms_per_step"] > 0 and 0.5 < d["warm_ms_per_step"] / d["cold_ms_per_step"] < 1.2
    assert "parity_gate" in d["config"] and d["scaling_quantity"] == "train_step.value" and d["world_size"] == 1
    assert d["channels_last"]["batches_rotated"] == 6
    v = r["valu"]
    assert "error" not in v, v
    assert 0.2 < v["valu_frac"] < 1.2 and v["shader_clock_mhz"] > 500 and v["tasks"] > 0 and v["pipe_cycles_per_row"] > 0
    c = d["cpu_baseline"]
    assert c["value"] > 0 and c["kind"] == "port" and c["cores"] >= 1 and c["unit"] == "images/s" and c["sample"]
    assert "--temporal --distil" in c["sample"]  # the same step as the headline, producer included
    t = d["train_step"]
    assert "error" not in t, t
    assert t["value"] > 0 and t["steps"] == 20 and "networks_fwd" in t["breakdown_ms"]
    assert d["eager_ms_per_step"] >= 0.5 * d["ms_per_step"]
    assert d["roofline_temporal"]["kernel_ms"] > 0


@pytest.mark.parametrize("extra,expect", [(["--mode", "dualrefine"], "DualRefine"), (["--width", "512"], "192x512"),
                                          (["--mode", "distil"], "--distil")],
                         ids=["dualrefine", "cityscapes_width", "distil"])
def test_other_bench_modes_print_a_line(extra, expect):
    d = _bench(["--gpus", "1", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--train-steps", "0", "--loss-blc", "--rotate", "2"] + extra)
    _contract(d, 2, 1)
    if extra == ["--mode", "distil"]:  # the reference's own command adds --loss_blc (README.md:22): timed as its own block
        b = d["loss_blc"]
        assert "error" not in b, b
        assert b["ms_per_step"] > 0 and b["launch"] == "eager" and b["w_ori"] > 0 and b["w_distil"] > 0
    assert expect in d["config"]["workload"], d["config"]["workload"]
    if extra[0] == "--mode" and extra[1] == "dualrefine":
        assert d["config"]["global_batch"] == 8 and abs(d["value"] - 8 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    if extra[0] == "--width":
        assert d["config"]["width"] == 512 and d["roofline"]["pixels_per_launch"] == 12 * 192 * 512


def test_two_ranks_over_gloo_print_the_multi_gpu_line():
    """the N>1 launch exactly as the driver starts it for N=2 -- `python bench.py --gpus 2` spawns the ranks itself -- but with
    gloo standing in for RCCL (MAL_BENCH_BACKEND=gloo: both ranks share this box's one card).  Covers everything of the
    multi-GPU line except the RCCL transport: rank spawn, max-over-ranks timing, and the shape of the N>1 line -- the SAME
    quantities as at N=1 (`value` = the loss path, `train_step` = RepDepth + loss path + the gradient pieces issued from inside
    the backward + Adam; no try/except around it at N>1: a failing collective fails the run)."""
    env = dict(os.environ, MAL_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--rotate", "2",
                        "--train-steps", "4"],
                       env=env, capture_output=True, text=True, timeout=1500, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.strip().startswith("{")][-1])
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 24 and d["scaling"] == "weak"
    # ONE quantity at every N: `value` is the loss path (here beside the stand-in 165 MB exchange after every step), the whole
    # training step sits in `train_step`, and `scaling_quantity` names the field the DP-scaling ratio is taken from
    assert "MAL loss path" in d["metric"] and "--temporal --distil" in d["config"]["workload"]
    assert "value_is" not in d and "loss_path" not in d and "cpu_baseline" not in d
    assert d["scaling_quantity"] == "train_step.value" and d["world_size"] == 2
    assert d["steps"] == 2 and d["warmup"] == 1
    assert abs(d["value"] - 24 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    assert d["breakdown_ms"]["grad_all_reduce"] > 0 and "error" not in d["exchange_overlapped"]
    t = d["train_step"]
    assert "error" not in t, t
    assert t["n_gpus"] == 2 and t["world_size"] == 2 and t["value"] > 0 and t["steps"] == 4
    assert "4 piece(s)" in t["exchange"] and "world size 2" in t["exchange"] and t["exchange_pieces"] == 4
    assert 0 <= t["pieces_issued_inside_backward"] <= 4
    assert abs(t["value"] - 24 / (t["ms_per_step"] * 1e-3)) <= 1e-6 * t["value"]
    assert d["roofline"]["kernel_ms"] > 0


def test_value_train_makes_the_training_step_the_line():
    """`--value train`: the line's value / ms_per_step are the whole training step's, timed over exactly --steps steps; the
    loss-path measurement moves to the `loss_path` side block"""
    d = _bench(["--gpus", "1", "--steps", "2", "--warmup", "1", "--value", "train", "--no-cpu-baseline", "--regime", "warm"])
    t = d["train_step"]
    assert "whole training step" in d["metric"] and d["value_is"].startswith("train_step")
    assert d["value"] == t["value"] and d["ms_per_step"] == t["ms_per_step"] and d["steps"] == 2 and t["steps"] == 2
    assert d["loss_path"]["value"] > 0 and d["scaling_quantity"] == "train_step.value"
    assert d["roofline"]["regime"] == "warm" and d["batches_rotated"] == 1
