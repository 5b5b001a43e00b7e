"""bench.py --gpus N exactly as the driver starts it — `python bench.py --gpus N --steps K --warmup W`, no launcher, no
WORLD_SIZE — on the one GPU a test box has: VIO_BENCH_ONE_DEVICE=1 puts both ranks on device 0 and stages the exchange
through pinned host memory over gloo (RCCL refuses two ranks on one device).  Everything else is the N > 1 path of the
bench: the parent spawns the rank processes before touching a GPU, the 200 000-landmark window of BASELINE.json configs[3]
is sharded over the ranks, one all-gather per iteration with the rank-ordered sum in the kernels, the unsharded window on
rank 0's GPU in the same run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(extra, env_extra=None, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=timeout)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    return p, (json.loads(lines[-1]) if lines else None)


def test_two_ranks_launched_as_the_driver_would():
    p, out = run_bench(["--gpus", "2", "--steps", "6", "--warmup", "2", "--replica-windows", "3", "--replica-landmarks", "4000", "--cpu-baseline-steps", "2"],
                       {"VIO_BENCH_ONE_DEVICE": "1"})
    assert p.returncode == 0, p.stderr[-3000:]
    assert out is not None, p.stdout[-2000:]
    assert out["n_gpus"] == 2 and out["steps"] == 6 and out["scaling"] == "strong"
    cfg = out["config"]
    assert cfg["landmarks_total"] == 200000 and cfg["landmarks_per_gpu"] == 100000 and cfg["observations_per_gpu"] == 400000
    assert "configs[3]" in cfg["workload"] and cfg["exchange"] == "hook_host" and cfg["all_ranks_on_one_device"] is True
    # one window's iterations per second, not ranks x iterations
    assert abs(out["value"] * out["ms_per_step"] - 1e3) <= 1e-6 * 1e3
    assert out["single_gpu_same_window_ms"] > 0 and out["speedup_vs_single_gpu_same_window"] > 0
    assert out["roofline"]["bound"] in ("mfma", "latency") and out["roofline"]["achieved"] > 0 and out["roofline"]["fp64"] is not None
    assert out["final_chi2"] > 0 and out["final_chi2"] == out["final_chi2"]
    # the CPU port on the whole shared window, and the regime that scales with GPUs: independent batches, no collective
    assert out["cpu_baseline"]["kind"] == "port" and out["cpu_baseline"]["value"] > 0 and "200000" in out["cpu_baseline"]["sample"]
    rep = out["replicas"]
    assert rep["windows_per_gpu"] == 3 and rep["scaling"] == "weak" and rep["window_iterations_per_s"] > 0
    assert cfg["rccl_ranks_seen"] == 0                  # (the host-staged exchange of the one-device form: no RCCL communicator)
    # the same window's single-GPU time measured inside the N > 1 run against a `--gpus 1 --landmarks 200000` run of its own
    # (two processes share the GPU in the first: the figure is taken by rank 0 behind a barrier, the other rank idle)
    p1, one = run_bench(["--gpus", "1", "--landmarks", "200000", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--batch", "0", "--no-per-frame", "--no-scale-projection"])
    assert p1.returncode == 0 and one["config"]["landmarks_total"] == 200000, p1.stderr[-2000:]
    assert abs(out["single_gpu_same_window_ms"] - one["ms_per_step"]) <= 0.25 * one["ms_per_step"], (out["single_gpu_same_window_ms"], one["ms_per_step"])


def test_eight_ranks_launched_as_the_driver_would():
    """The driver's own 8-GPU command — `python bench.py --gpus 8 --steps K --warmup W` — with all eight rank processes on the one device of a
    test box: eight processes each synthesising the 200 000-landmark window, the rendezvous, 25 000 landmarks per rank (one round of
    k_linearize workgroups), the exchange of eight 24 KB slabs summed in rank order, cpu_baseline on the shared window, the `replicas`
    block.  The first real 8-GPU launch is then not also the first 8-process launch."""
    p, out = run_bench(["--gpus", "8", "--steps", "4", "--warmup", "1", "--replica-windows", "1", "--replica-landmarks", "2000", "--cpu-baseline-steps", "1"],
                       {"VIO_BENCH_ONE_DEVICE": "1"}, timeout=1500)
    assert p.returncode == 0, p.stderr[-3000:]
    assert out is not None, p.stdout[-2000:]
    assert out["n_gpus"] == 8 and out["steps"] == 4 and out["scaling"] == "strong"
    cfg = out["config"]
    assert cfg["landmarks_total"] == 200000 and cfg["landmarks_per_gpu"] == 25000 and cfg["observations_per_gpu"] == 100000
    assert cfg["exchange"] == "hook_host" and cfg["all_ranks_on_one_device"] is True
    assert abs(out["value"] * out["ms_per_step"] - 1e3) <= 1e-6 * 1e3
    assert out["final_chi2"] > 0 and out["final_chi2"] == out["final_chi2"]
    assert out["single_gpu_same_window_ms"] > 0
    assert out["cpu_baseline"]["value"] > 0 and "200000" in out["cpu_baseline"]["sample"]
    assert out["replicas"]["windows_per_gpu"] == 1 and out["replicas"]["window_iterations_per_s"] > 0


def test_single_gpu_line_carries_the_measured_scale_projection():
    """N = 1: the sharded launch sequence on rank 0's 200 000 / N-landmark shard, N = 2, 4, 8, with the library's RCCL all-gather on a
    one-rank communicator, against the unsharded window in the same run."""
    p, out = run_bench(["--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--batch", "0", "--no-per-frame"])
    assert p.returncode == 0, p.stderr[-3000:]
    # the contract: ONE JSON line on stdout — RCCL's version banner (a communicator is created for the projection) and anything else a library
    # prints there must not be on it
    assert [ln for ln in p.stdout.splitlines() if ln.strip()] == [ln for ln in p.stdout.splitlines() if ln.startswith("{")] and p.stdout.count("\n") == 1, p.stdout[:600]
    sp = out["scale_projection"]
    if "error" in sp:
        # (seen once in some twenty runs of this file, not reproduced in the eleven that followed: the block creates and destroys three one-rank
        #  RCCL communicators in a row; one more attempt before the run counts as failed, with the first attempt's message on the record)
        print("scale_projection failed once:", sp)
        p, out = run_bench(["--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--batch", "0", "--no-per-frame"])
        assert p.returncode == 0, p.stderr[-3000:]
        sp = out["scale_projection"]
    assert "error" not in sp, sp
    assert sp["unsharded_ms_per_iteration"] > 0
    prev = sp["unsharded_ms_per_iteration"]
    for n_sh in ("2", "4", "8"):
        e = sp["shards"][n_sh]
        assert e["landmarks_per_gpu"] == 200000 // int(n_sh) and e["exchange"] == "native" and e["rccl_ranks_seen"] == 1
        assert 0 < e["ms_per_iteration_per_shard"] < prev * 1.05          # a smaller shard is not slower
        assert abs(e["projected_speedup"] - sp["unsharded_ms_per_iteration"] / e["ms_per_iteration_per_shard"]) < 0.01
        prev = e["ms_per_iteration_per_shard"]
    assert sp["shards"]["8"]["projected_speedup"] < 4.0                   # Amdahl: the replicated pose solve


def test_a_rank_that_dies_ends_the_whole_run():
    import time
    t0 = time.time()
    p, out = run_bench(["--gpus", "2", "--steps", "4", "--warmup", "1"], {"VIO_BENCH_ONE_DEVICE": "1", "VIO_BENCH_FAIL_RANK": "1"}, timeout=120)
    assert p.returncode == 3 and out is None and time.time() - t0 < 60.0


def test_single_gpu_line_is_the_headline_window():
    p, out = run_bench(["--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--batch", "0", "--no-per-frame", "--no-scale-projection"])
    assert p.returncode == 0, p.stderr[-3000:]
    assert out["n_gpus"] == 1 and out["scaling"] == "weak"
    assert out["config"]["landmarks_total"] == 20000 and out["config"]["observations_per_gpu"] == 80000
    assert out["single_gpu_same_window_ms"] is None
    assert abs(out["value"] * out["ms_per_step"] - 1e3) <= 1e-6 * 1e3
