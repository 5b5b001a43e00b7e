#!/usr/bin/env python3
"""bench.py — GN iterations/s of the MI355X sliding-window VIO backend on BASELINE.json's headline config.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--landmarks L] [--landmark-type invdepth|xyz]

One "step" = one Gauss-Newton iteration of the hot path: linearise all reprojection
+ IMU factors, reduce the landmark Schur complement, add the prior, damped pivoted LDLT of the 171x171 pose
system, back-substitute the landmarks, update every state, re-evaluate chi2 — `vio_gn_iteration` of
include/vio_backend.h, enqueued back to back with no host round trip.  Inputs are resident in HBM before the
timed region starts.

N = 1: the 11-frame / 20 000-landmark / 80 000-observation synthetic window (BASELINE.json configs[2]).
N > 1: one process per GPU (torch.distributed, backend nccl == RCCL, for the rendezvous, the barriers and the
       128-byte communicator id); every rank holds a 20 000-landmark shard of an (N x 20 000)-landmark window;
       per iteration the library itself issues one RCCL all-reduce of the 72x72 reduced visual system and one
       of the 2 step scalars on its stream.  Weak scaling: `value` counts
       shard-iterations per second over all ranks.

The JSON line also carries
  roofline      achieved = algorithmic bytes per launch / measured launch duration of the dominant kernel
                (HIP event pairs on the library's stream around about ten launches spread over the timed steps),
                peak = 8 TB/s HBM
  cpu_baseline  the oracle's (oracle/vio_oracle.c, plain C, 1 thread) GN iteration on the same window,
                timed on this box's host cores on a bounded sample (rank 0, N = 1 only)
  per_frame     what surrounds the inner iteration once per frame: vio_set_*, plan + upload + first linearisation,
                Solve(10), MargOldFrame — host wall clock, for the HIP library and for the CPU port
"""
import argparse
import importlib.util
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "visual-inertial-odometry_amd")


def load_package():
    if "vio_amd" in sys.modules:
        return sys.modules["vio_amd"]
    spec = importlib.util.spec_from_file_location(
        "vio_amd", os.path.join(PKG_DIR, "__init__.py"), submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["vio_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


# algorithmic bytes of one launch of each kernel, N landmarks / M observations on this GPU
def kernel_algorithmic_bytes(name, n, m, xyz=False):
    if xyz:     # per observation (x, y) fp64 + 2 int32 of indices; per landmark 3 fp64 read, 3 fp64 written (+ 3 of delta)
        b_state, b_imu = 1464, 10 * (10 + 225 + 225 + 6 + 1) * 8
        b_sys = (171 * 171 + 171) * 8
        return {"k_linearize": 24 * m + 48 * n + b_state + b_imu, "k_reduce": 78 * 36 * 8 * 2, "k_assemble": 2 * b_sys,
                "k_pose_solve": 2 * b_sys + b_state + 171 * 8, "k_backsub": 24 * m + 72 * n + b_state, "k_lm_decide": 4096}[name]
    b_state = 1464
    b_imu = 10 * (10 + 225 + 225 + 6 + 1) * 8
    b_sys = (171 * 171 + 171) * 8
    return {
        # reads: 2 (x,y) fp64 per observation + per-landmark host (x,y) + inverse depth + 3 int32 of indices per
        # observation (44 B/obs + 8 B/landmark); writes h_ll/b_l per landmark
        "k_linearize": 44 * m + 16 * n + b_state + b_imu,
        "k_reduce": 78 * 36 * 8 * 2,
        "k_assemble": 2 * b_sys,
        "k_pose_solve": 2 * b_sys + b_state + 171 * 8,
        # reads pts (44 B/obs), inverse depth + writes the trial inverse depth and delta per landmark
        "k_backsub": 44 * m + 24 * n + b_state,
        "k_lm_decide": 4096,
    }[name]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--landmarks", type=int, default=20000, help="landmarks per GPU")
    ap.add_argument("--obs-per-landmark", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prior", action="store_true", help="first-window case: no marginalisation prior")
    ap.add_argument("--landmark-type", choices=("invdepth", "xyz"), default="invdepth",
                    help="invdepth: VertexInverseDepth + EdgeReprojection (what Estimator builds, the headline); "
                         "xyz: VertexPointXYZ + EdgeReprojectionXYZ (3x3 landmark blocks)")
    ap.add_argument("--batch", type=int, default=64, help="windows of the batched block (vio_batch_gn_iteration); 0 skips it")
    ap.add_argument("--no-per-frame", action="store_true", help="skip the per-frame cost block (set / plan+upload / Solve(10) / marginalise)")
    ap.add_argument("--cpu-baseline-steps", type=int, default=0, help="0 = sized for about 10-20 s")
    args = ap.parse_args()

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("VIO_BENCH_ONE_DEVICE") == "1":      # diagnostic: all ranks on device 0 (if RCCL lets them)
        local_rank = 0
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node %d bench.py --gpus %d ..."
                             % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback exists)")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    vio = load_package()
    hip = vio.load_hip()        # raises if csrc/libvio_hip.so is missing: no fallback path

    n_per_gpu, k_obs = args.landmarks, args.obs_per_landmark
    xyz = args.landmark_type == "xyz"
    make = vio.synth.make_window_xyz if xyz else vio.synth.make_window
    full = make(n_per_gpu * world, seed=42, obs_per_landmark=k_obs)
    if not args.no_prior:
        # the steady-state window carries a marginalisation prior ("produced by running one MargOldFrame
        # on a preceding window"; its 235 KB are part of B_win): solve the window one frame earlier, marginalise its
        # oldest frame (rank 0; the others receive the same bytes)
        prior = None
        if rank == 0:
            wp = vio.synth.make_window(300, seed=41, t0=0.9, obs_per_landmark=k_obs)
            cp = hip.context(device=local_rank)
            cp.load(wp)
            cp.solve(10)
            prior = cp.marginalize(vio.MARG_OLD)
            del cp
        if dist is not None:
            box = [prior]
            dist.broadcast_object_list(box, src=0)
            prior = box[0]
        full.prior = prior

    # N > 1: the library all-reduces with RCCL itself on its own stream (VIO_EXCHANGE=hook selects the
    # torch.distributed hook instead, which has to share torch's current stream)
    kw = dict(device=local_rank)       # (with VIO_EXCHANGE=hook, ShardedBackend puts the library and the collective on one torch stream)
    force = os.environ.get("VIO_BENCH_FORCE_EXCHANGE") == "1"      # diagnostic: run the sharded kernel sequence on one rank
    if force and dist is None and os.environ.get("VIO_EXCHANGE", "native") == "hook":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", local_rank))
    sb = vio.sharded.ShardedBackend(hip, full, rank, world, dist=dist, torch_device="cuda", ctx_kwargs=kw, force_hook=force)
    ctx, w = sb.ctx, sb.shard
    n, m = w.n_landmarks, w.n_observations

    # lambda of the reference's LM start (ComputeLambdaInitLM): identical on every rank after the exchange
    ctx.linearize()
    _, lam = ctx.init_lm()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        ctx.gn_iteration(lam)
    ctx.synchronize()

    # which kernel dominates?  one short profiled pass per kernel (outside the timed region)
    per_kernel = {}
    for kid, name in enumerate(hip.KERNELS):
        ctx.profile_begin(kid)
        for _ in range(10):
            ctx.gn_iteration(lam)
        ms, cnt = ctx.profile_end()
        per_kernel[name] = ms / max(cnt, 1)
    dominant = max(per_kernel, key=per_kernel.get)

    # timed region: exactly K steps; an event pair around every `stride`-th launch of the dominant kernel — about ten pairs
    # over the region, never fewer than every 8th launch (a hipEventRecord drains the stream, ~7 us each: bracketing every
    # launch would add ~15 % to the step it is meant to observe, every 8th still 2.5 %)
    stride = max(8, args.steps // 10)
    ctx.profile_begin_sampled(hip.KERNELS.index(dominant), stride)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ctx.gn_iteration(lam)
    ctx.synchronize()
    barrier()
    t1 = time.perf_counter()
    dom_ms, dom_cnt = ctx.profile_end()
    elapsed = t1 - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    chi2 = ctx.chi2()
    ms_per_step = elapsed * 1e3 / args.steps
    value = world * args.steps / elapsed

    dom_launch_s = (dom_ms / max(dom_cnt, 1)) * 1e-3
    alg_bytes = kernel_algorithmic_bytes(dominant, n, m, xyz)
    achieved = alg_bytes / dom_launch_s / 1e9 if dom_launch_s > 0 else 0.0
    # HBM bytes per launch: NOT measured in this run (PMC counters need rocprofv3 around the process): the figure of the
    # committed counter pass over this same command (profiles/traffic.json <- tools/summarize_profile.py), labelled as such
    traffic, traffic_source = None, None
    tr_path = os.path.join(ROOT, "profiles", "traffic_xyz.json" if xyz else "traffic.json")
    if os.path.exists(tr_path):
        try:
            traffic = json.load(open(tr_path)).get(dominant, {}).get("hbm_bytes_per_launch")
            traffic_source = "committed rocprofv3 --pmc pass (profiles/%s), not this run" % os.path.basename(tr_path)
        except Exception:
            traffic = None
    it_bytes = (24 * m + 72 * n + 280000) if xyz else vio.synth.algorithmic_bytes(n, m)
    roofline = {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 3), "peak": 8000.0, "unit": "GB/s",
                "frac": achieved / 8000.0, "traffic": traffic, "traffic_source": traffic_source,
                "algorithmic_bytes_per_launch": alg_bytes, "launch_us": round(dom_launch_s * 1e6, 3),
                "launch_us_method": "HIP event pairs on the library's stream around every %d-th launch of the timed steps " % stride +
                                    "(includes the event's own drain, ~8 % above rocprofv3's kernel duration)",
                "iteration_algorithmic_bytes": it_bytes,
                "iteration_achieved_GBps": round(it_bytes / (ms_per_step * 1e-3) / 1e9, 3),
                "kernel_us_event_bracketed": {k: round(v * 1e3, 3) for k, v in per_kernel.items()}}

    # ---- what one frame costs around the inner iteration (Estimator::backendOptimization, estimator.cpp:1075-1141):
    #      vio_set_* of a fresh window, plan + upload + first linearisation, Solve(10), MargOldFrame — host wall clock
    per_frame = None
    if rank == 0 and world == 1 and not args.no_per_frame:
        other = make(n_per_gpu, seed=43, obs_per_landmark=k_obs)
        other.prior = full.prior

        def frame_costs(lib, reps):
            c = lib.context(**({"device": local_rank} if lib is hip else {}))
            acc = {"set_ms": 0.0, "plan_upload_linearize_ms": 0.0, "solve10_ms": 0.0, "marginalize_ms": 0.0}
            iters = 0
            for r in range(reps + 1):
                t0 = time.perf_counter()
                c.load(full if r % 2 == 0 else other)      # a frame never repeats the one before: the library skips inputs it already holds
                t1 = time.perf_counter()
                c.linearize()
                if lib is hip:
                    c.synchronize()
                t2 = time.perf_counter()
                rep = c.solve(10)
                t3 = time.perf_counter()
                t4 = t3
                if not xyz:
                    c.marginalize(vio.MARG_OLD)
                    t4 = time.perf_counter()
                if r == 0:
                    continue                # first pass: allocations, first touches
                acc["set_ms"] += (t1 - t0) * 1e3; acc["plan_upload_linearize_ms"] += (t2 - t1) * 1e3
                acc["solve10_ms"] += (t3 - t2) * 1e3; acc["marginalize_ms"] += (t4 - t3) * 1e3
                iters = rep.iterations
            out = {k: round(v / reps, 4) for k, v in acc.items()}
            out["frame_ms"] = round(sum(out.values()), 4)
            out["solve10_iterations"] = iters
            if xyz:
                out["marginalize_ms"] = None      # MargOldFrame is not defined for XYZ landmarks (include/vio_backend.h)
            return out
        per_frame = {"gpu": frame_costs(hip, 5),
                     "note": "host wall clock per call on the bench window; set = vio_set_window/landmarks/observations/imu/prior "
                             "(host copies), plan_upload_linearize = pattern grouping + H2D + first linearisation, marginalize = "
                             "MargOldFrame: GPU assembly + Schur, 171x171 D2H, eigen-decomposition tail on one host thread"}

    # ---- B independent windows per launch (vio_batch_gn_iteration): the regime in which the device is full.  Same window
    #      size as the headline, different seeds; reported beside the single-window line, never instead of it
    batched = None
    if rank == 0 and world == 1 and args.batch > 0:
        B = args.batch
        lead = hip.context(device=local_rank)
        members = [lead] + [hip.context(device=local_rank, stream=lead.get_stream()) for _ in range(B - 1)]
        wbs = []
        for i, cb in enumerate(members):
            wb = (vio.synth.make_window_xyz if xyz else vio.synth.make_window)(n_per_gpu, seed=100 + i, obs_per_landmark=k_obs)
            wb.prior = full.prior
            cb.load(wb)
            wbs.append(wb)
        for _ in range(5):
            hip.batch_gn_iteration(members, lam)
        lead.synchronize()
        bsteps = max(20, args.steps // 4)
        tb = time.perf_counter()
        for _ in range(bsteps):
            hip.batch_gn_iteration(members, lam)
        lead.synchronize()
        tb = time.perf_counter() - tb
        bytes_it = vio.synth.algorithmic_bytes(n, m)
        batched = {"windows": B, "steps": bsteps, "ms_per_batch_iteration": tb * 1e3 / bsteps,
                   "window_iterations_per_s": B * bsteps / tb, "us_per_window_iteration": tb * 1e6 / (bsteps * B),
                   "algorithmic_GBps": round(B * bytes_it * bsteps / tb / 1e9, 2), "hbm_frac": B * bytes_it * bsteps / tb / 8e12,
                   "final_chi2_window0": lead.chi2(),
                   "note": "B independent 20k-landmark windows (seeds 100..), one launch per kernel for all of them (grid.y = window); "
                           "bit-identical to B separate vio_gn_iteration runs (tests/test_gpu_batch.py)"}
        # Problem::Solve(10) of the same B windows in one batched call (vio_batch_solve), from their initial states; the uploads
        # in front of it are not timed (per_frame has those)
        ts, its = [], 0
        for _ in range(3):
            for cb, wb in zip(members, wbs):
                cb.load(wb)
                cb.linearize()
            lead.synchronize()
            tb0 = time.perf_counter()
            reps_b = hip.batch_solve(members, 10)
            ts.append(time.perf_counter() - tb0)
            its = sum(r.iterations for r in reps_b) / len(reps_b)
        batched["solve10_ms_per_batch"] = min(ts) * 1e3
        batched["solve10_ms_per_window"] = min(ts) * 1e3 / B
        batched["solve10_mean_iterations"] = its
        del cb, members, lead          # (members first: they run on the leader's stream)

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s"])
        orc = vio.VioLib(os.path.join(ROOT, "oracle", "liboracle.so"), "vioo_")
        if per_frame is not None:
            per_frame["cpu_port_1_thread"] = frame_costs(orc, 1)
        co = orc.context()
        co.load(full)
        co.gn_iteration(lam)        # warm-up / first touch
        t = time.perf_counter()
        co.gn_iteration(lam)
        one = time.perf_counter() - t
        steps = args.cpu_baseline_steps or max(3, min(200, int(12.0 / max(one, 1e-3))))
        t = time.perf_counter()
        for _ in range(steps):
            co.gn_iteration(lam)
        dt = time.perf_counter() - t
        cpu_baseline = {"value": steps / dt, "unit": "GN iter/s", "cores": 1, "kind": "port",
                        "ms_per_iter": dt * 1e3 / steps,
                        "sample": "%d GN iterations of the same %d-landmark / %d-observation window, oracle/vio_oracle.c "
                                  "(plain C, -O2, 1 thread)" % (steps, n, m)}

    # the same port on all host cores (OpenMP over landmarks, oracle/liboracle_omp.so; SURVEY.md 8d asks for both figures)
    cpu_baseline_all_cores = None
    if cpu_baseline is not None and not xyz:       # (the OpenMP build parallelises the inverse-depth path only)
        try:
            import subprocess
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "omp"])
            omp = vio.VioLib(os.path.join(ROOT, "oracle", "liboracle_omp.so"), "vioo_")
            cm = omp.context()
            cm.load(full)
            cm.gn_iteration(lam)
            t = time.perf_counter()
            cm.gn_iteration(lam)
            one = time.perf_counter() - t
            steps = max(3, min(300, int(6.0 / max(one, 1e-3))))
            t = time.perf_counter()
            for _ in range(steps):
                cm.gn_iteration(lam)
            dt = time.perf_counter() - t
            navail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
            ncores = min(16, navail)           # oracle_threads(): one window's work saturates at about 16 threads
            cpu_baseline_all_cores = {"value": steps / dt, "unit": "GN iter/s", "cores": ncores, "kind": "port",
                                      "ms_per_iter": dt * 1e3 / steps,
                                      "sample": "%d GN iterations of the same window, oracle/vio_oracle.c built with -fopenmp "
                                                "(landmark-parallel linearisation, Schur terms, back-substitution and chi2; "
                                                "the 171x171 LDLT stays serial), %d threads of the %d available" % (steps, ncores, navail)}
        except Exception as exc:      # a reported extra, never a reason to lose the line
            cpu_baseline_all_cores = {"error": str(exc)}

    # the reference's own backend (compiled from its sources where they lie, oracle/_ref, by the recipe in oracle/Makefile;
    # present when the repo was built where /root/reference exists): its dense (171+N)^2 solver needs 18 s and 13 GB per
    # iteration at N = 20 000, so it is timed on a 2 000-landmark window of the same generator and reported beside the port
    cpu_reference = None
    ref_so = os.path.join(ROOT, "oracle", "_ref", "libvio_ref.so")
    if cpu_baseline is not None and os.path.exists(ref_so):
        try:
            rl = vio.VioLib(ref_so, "vior_")
            wr = make(2000, seed=42, obs_per_landmark=k_obs)
            cr = rl.context()
            cr.load(wr)
            cr.linearize()
            _, lam_r = cr.init_lm()
            cr.gn_iteration(lam_r)
            t = time.perf_counter()
            nref = 5
            for _ in range(nref):
                cr.gn_iteration(lam_r)
            dt = time.perf_counter() - t
            co2 = orc.context()
            co2.load(wr)
            co2.gn_iteration(lam_r)
            t = time.perf_counter()
            for _ in range(50):
                co2.gn_iteration(lam_r)
            dto = (time.perf_counter() - t) / 50
            cpu_reference = {"value": nref / dt, "unit": "GN iter/s", "cores": 1, "kind": "reference",
                             "ms_per_iter": dt * 1e3 / nref, "port_ms_per_iter_same_window": dto * 1e3,
                             "sample": "%d GN iterations of a 2000-landmark / %d-observation window (the reference's dense "
                                       "solver is O(N^2): 18 s per iteration at N = 20000, BASELINE.md section 2), "
                                       "oracle/_ref/libvio_ref.so, 1 thread" % (nref, wr.n_observations)}
        except Exception as exc:       # the reference library is optional test infrastructure
            cpu_reference = {"error": str(exc)}

    if rank == 0:
        out = {
            "metric": "GN iterations/s, 11-frame (10-keyframe) window, 20k landmarks per GPU",
            "value": value, "unit": "GN iter/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "synthetic 11-frame VIO window (SURVEY.md 8d): %d landmarks x %d observations per GPU, "
                                   "10 IMU factors, %s, Cauchy loss, extrinsic fixed, fixed-lambda GN iteration, %s"
                                   % (n_per_gpu, k_obs + (1 if xyz else 0), "no prior" if args.no_prior else "marginalisation prior of the preceding window",
                                      "XYZ landmarks (VertexPointXYZ, 3x3 blocks)" if xyz else "inverse-depth landmarks"),
                       "landmark_type": args.landmark_type,
                       "landmarks_per_gpu": n_per_gpu, "observations_per_gpu": m, "landmarks_total": n_per_gpu * world,
                       "lambda": lam, "parallelism": "landmark-sharded x%d, all-reduce of the 72x72 reduced system" % world
                       if world > 1 else "single GPU", "exchange": sb.exchange},
            "final_chi2": chi2,
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
            "per_frame": per_frame,
            "batched": batched,
            "cpu_baseline_all_cores": cpu_baseline_all_cores,
            "cpu_reference": cpu_reference,
        }
        import ctypes
        ctypes.CDLL(None).fflush(None)      # RCCL's version banner sits in C stdio's buffer: keep the JSON line last
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
