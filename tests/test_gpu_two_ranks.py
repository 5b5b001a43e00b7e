"""The N = 2 protocol of the HIP library with two REAL ranks: two processes, each with its own
context and its own shard of the landmarks, both on the one GPU a test box has.  RCCL refuses two ranks on one device, so
the exchange is the library's hook with the 24 KB staged through pinned host memory and all-reduced with gloo
(`exchange="hook_host"`, visual-inertial-odometry_amd/sharded.py) — the kernels around it (k_reduce -> exchange ->
k_assemble, k_step_sum -> exchange -> k_lm_decide, the deferred sums of the GN loop, the sharded MargOldFrame) are exactly
the ones the RCCL path runs, and here the sums are not identities.

Checked: LM solve(10), five fixed-lambda GN iterations and MargOldFrame of the two-rank run equal the unsharded HIP run to
the tolerances of tests/test_distributed_cpu.py, and the two ranks agree bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make(vio, kind, n, ragged):
    # (tools/fuzz_two_ranks.py varies the seed and adds a prior through the environment, which the spawned ranks inherit)
    seed = int(os.environ.get("VIO_TWO_RANK_SEED", "21"))
    w = vio.synth.make_window_xyz(n, seed=seed, ragged=ragged) if kind == "xyz" else vio.synth.make_window(n, seed=seed, ragged=ragged)
    if os.environ.get("VIO_TWO_RANK_PRIOR") == "1":
        p = np.random.RandomState(seed).normal(size=(156, 40))
        H = p @ p.T * 1e3                           # any symmetric positive semi-definite prior will do for the protocol
        ev, V = np.linalg.eigh(H)
        keep = ev > 1e-8 * ev.max()
        J = (V[:, keep] / np.sqrt(ev[keep])).T
        jt = np.zeros((156, 156))
        jt[:J.shape[0]] = J
        b = H @ np.random.RandomState(seed + 1).normal(scale=1e-3, size=156)
        w.prior = dict(H=H, b=b, err=-(jt @ b), jt_inv=jt)
    return w


def _worker(rank, world, port, out_dir, kind, n, ragged):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from conftest import load_package
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    vio = load_package()
    hip = vio.load_hip()
    w = _make(vio, kind, n, ragged)
    sb = vio.sharded.ShardedBackend(hip, w, rank, world, dist=dist, torch_device="cuda", exchange="hook_host")
    out = {}
    sb.ctx.linearize()
    out["chi0"], out["lam0"] = sb.ctx.init_lm()
    out["Hs"], out["bs"] = sb.ctx.get_schur_system()
    rep = sb.solve(10)
    out["poses"], out["sb"], _ = sb.ctx.get_window()
    local = sb.ctx.get_landmarks_xyz() if kind == "xyz" else sb.ctx.get_landmarks()
    parts = [None] * world
    dist.all_gather_object(parts, local)
    out["lms"] = np.concatenate(parts)
    out["final_chi2"], out["iterations"], out["trials"] = rep.final_chi2, rep.iterations, rep.trials
    if kind != "xyz":
        m = sb.marginalize(vio.capi.MARG_OLD)
        out.update(marg_H=m["H"], marg_b=m["b"], marg_err=m["err"], marg_jt=m["jt_inv"])
    # GN loop from the initial state again: the step scalars ride with the next linearisation's exchange
    sb.ctx.load(sb.shard)
    for _ in range(5):
        sb.gn_iteration(float(out["lam0"]))
    out["gn_poses"], out["gn_sb"], _ = sb.ctx.get_window()
    out["gn_chi2"] = sb.ctx.chi2()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **out)
    dist.destroy_process_group()


@pytest.mark.parametrize("kind,n,ragged,world", [("invdepth", 900, True, 2), ("invdepth", 5000, False, 2), ("xyz", 700, True, 2), ("invdepth", 3000, True, 4)])
def test_two_hip_ranks_equal_the_unsharded_run(vio, hip_lib, tmp_path, kind, n, ragged, world):
    """(world 4: the rank-ordered sums of the kernels over four gathered slabs — what N = 4 / 8 GPUs run — still on the one device)"""
    run_ranks_case(vio, hip_lib, tmp_path, kind, n, ragged, world)


def run_ranks_case(vio, hip_lib, tmp_path, kind, n, ragged, world=2):
    import torch.multiprocessing as mp
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), kind, n, ragged), nprocs=world, join=True)
    ranks = [np.load(tmp_path / ("rank%d.npz" % r)) for r in range(world)]
    r0 = ranks[0]
    w = _make(vio, kind, n, ragged)
    ctx = hip_lib.context()
    ctx.load(w)
    ctx.linearize()
    chi0, lam0 = ctx.init_lm()
    Hs, bs = ctx.get_schur_system()
    rep = ctx.solve(10)
    poses, sbias, _ = ctx.get_window()
    lms = ctx.get_landmarks_xyz() if kind == "xyz" else ctx.get_landmarks()
    marg = ctx.marginalize(vio.capi.MARG_OLD) if kind != "xyz" else None
    ctx.load(w)
    for _ in range(5):
        ctx.gn_iteration(lam0)
    gp, gs, _ = ctx.get_window()
    gchi = ctx.chi2()
    for r in ranks:
        assert abs(float(r["chi0"]) - chi0) <= 1e-12 * abs(chi0) and float(r["lam0"]) == lam0
        d = np.sqrt(np.abs(np.diag(Hs)) + 1e-300)
        assert (np.abs(r["Hs"] - Hs) / np.outer(d, d)).max() <= 1e-12
        assert int(r["iterations"]) == rep.iterations and int(r["trials"]) == rep.trials, (int(r["iterations"]), rep.iterations, int(r["trials"]), rep.trials, float(r["final_chi2"]), rep.final_chi2)
        assert abs(float(r["final_chi2"]) - rep.final_chi2) <= 1e-7 * rep.final_chi2
        assert np.abs(r["poses"] - poses).max() <= 1e-7 and np.abs(r["sb"] - sbias).max() <= 1e-6
        assert np.abs(r["lms"] - lms).max() <= 1e-7
        assert np.abs(r["gn_poses"] - gp).max() <= 1e-7 and np.abs(r["gn_sb"] - gs).max() <= 1e-6
        assert abs(float(r["gn_chi2"]) - gchi) <= 1e-7 * gchi
        if marg is not None:
            mscale = np.abs(marg["H"]).max()
            assert np.abs(r["marg_H"] - marg["H"]).max() <= 2e-5 * mscale
            evs, evr = np.linalg.eigvalsh(r["marg_H"]), np.linalg.eigvalsh(marg["H"])
            assert np.abs(evs - evr).max() <= 2e-5 * evr.max()
            assert np.abs(r["marg_b"] - marg["b"]).max() <= 1e-6 * max(1.0, np.abs(marg["b"]).max())
    for r1 in ranks[1:]:
        for k in r0.files:          # every rank holds the identical reduced system and takes the identical steps
            np.testing.assert_array_equal(r0[k], r1[k], err_msg=k)
