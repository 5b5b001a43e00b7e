"""CPU tests of the boundary and the host logic: the C-ABI library loads and exports every symbol the header
declares, the synthetic generator produces what the ABI takes, sharding partitions the landmarks."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import vio_testutil as tu
from conftest import ROOT


def header_functions():
    txt = open(os.path.join(ROOT, "include", "vio_backend.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"#ifdef VIO_DEBUG_ENTRY_POINTS.*?#endif", "", txt, flags=re.S)      # not part of the product ABI
    names = set(re.findall(r"\b(vio_[a-z0-9_]+)\s*\(", txt))
    names.discard("vio_exchange_fn")
    return sorted(names)


def test_header_declares_the_reference_call_sequence():
    fns = header_functions()
    for need in ("vio_create", "vio_set_window", "vio_set_landmarks", "vio_set_observations", "vio_set_imu",
                 "vio_set_prior", "vio_solve", "vio_marginalize", "vio_get_window", "vio_get_prior", "vio_destroy"):
        assert need in fns


def test_header_is_plain_c(tmp_path):
    """The boundary is a C ABI: include/vio_backend.h compiles as C99 and as C++11, pedantic, with nothing but the standard headers."""
    import subprocess
    src = tmp_path / "hdr.c"
    src.write_text('#include "vio_backend.h"\nint main(void) { vio_config c; vio_solve_report r; (void)c; (void)r; return 0; }\n')
    inc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", inc, "-c", str(src), "-o", str(tmp_path / "a.o")])
    subprocess.check_call(["g++", "-std=c++11", "-Wall", "-Wextra", "-Werror", "-pedantic", "-I", inc, "-x", "c++", "-c", str(src), "-o", str(tmp_path / "b.o")])


def test_hip_library_exports_every_declared_symbol(vio):
    """No compute call here (there is no GPU in the CPU tier): only that the product library exists, loads and
    resolves each prototype of include/vio_backend.h."""
    lib = vio.load_hip().dll
    missing = [f for f in header_functions() if not hasattr(lib, f)]
    assert not missing, missing


def test_hip_library_exports_nothing_but_the_abi(vio):
    """-fvisibility=hidden: the dynamic symbol table of the product library holds the functions include/vio_backend.h declares and nothing
    else of the library's own (no vio_launch_* / vio_chain_* internals, no diagnostic entry points)."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", vio.HIP_LIB], capture_output=True, text=True, check=True).stdout
    exported = {l.split()[-1] for l in out.splitlines() if l.split() and l.split()[-2] in ("T", "W", "B", "D", "V")}
    ours = {n for n in exported if "vio" in n.lower() or n.startswith("k_") or "lin_" in n}
    extra = sorted(ours - set(header_functions()))
    assert not extra, extra
    assert "vio_debug_chain_solve" not in exported


def test_host_code_of_the_product_library_has_no_fused_multiply_add(vio):
    """csrc/host_dense.cpp promises the same bits from its baseline, AVX2 and AVX-512 paths (the priors of a stream must not depend on the
    host they were computed on): none of them may contract a product with the sum behind it.  hipcc compiles the file with -ffp-contract=fast,
    so the promise is checked where it can break — in the library's x86 code: no vfmadd / vfmsub / vfnmadd instruction at all."""
    import shutil
    import subprocess
    if shutil.which("objdump") is None:
        pytest.skip("no objdump here")
    out = subprocess.run(["objdump", "-d", "--no-show-raw-insn", vio.HIP_LIB], capture_output=True, text=True, check=True).stdout
    fused = [l for l in out.splitlines() if "\tvfmadd" in l or "\tvfmsub" in l or "\tvfnmadd" in l or "\tvfnmsub" in l]
    assert not fused, fused[:5]


def test_abi_version(vio):
    import ctypes as C
    f = vio.load_hip().dll.vio_abi_version
    f.restype = C.c_int32
    txt = open(os.path.join(ROOT, "include", "vio_backend.h")).read()
    assert f() == int(re.search(r"#define VIO_ABI_VERSION (\d+)", txt).group(1)) >= 4


def test_hip_library_fails_loudly_without_a_gpu(vio):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    lib = vio.load_hip()
    with pytest.raises(vio.VioError) as e:
        lib.context()
    assert e.value.status in (-6, -2)      # VIO_ERR_NO_DEVICE / VIO_ERR_HIP, never a silent CPU path


def test_oracle_exports_the_same_surface(vio, oracle_lib):
    for f in header_functions():
        if f == "vio_preintegrate":
            assert oracle_lib.has("preintegrate_abi")
            continue
        if f in ("vio_profile_begin", "vio_profile_begin_sampled", "vio_profile_end", "vio_kernel_name",
                 "vio_comm_unique_id", "vio_comm_init", "vio_comm_destroy", "vio_comm_info", "vio_get_stream", "vio_batch_gn_iteration", "vio_batch_solve",
                 "vio_get_host_timing", "vio_set_solve_order", "vio_get_solve_order", "vio_abi_version"):
            continue        # measurement hooks, the native RCCL exchange, streams, batched launches and the choice of the GPU
                            # solver's elimination order exist on the HIP library only
        assert oracle_lib.has(f[len("vio_"):]), f


def test_synthetic_window_shapes(vio):
    w = vio.synth.make_window(70, seed=3)
    assert w.poses.shape == (11, 7) and w.speed_bias.shape == (11, 9) and w.ext.shape == (7,)
    assert w.inv_depth.shape == (70,) and w.lm.shape == (280,) and w.pts_j.shape == (280, 2)
    assert len(w.preint) == 10 and abs(w.preint[0]["sum_dt"] - 0.1) < 1e-12
    assert np.all(w.host != w.target) and w.target.max() <= 10
    np.testing.assert_allclose(np.linalg.norm(w.poses[:, 3:7], axis=1), 1.0, atol=1e-12)
    # observations are grouped by landmark, as estimator.cpp:975-1016 emits them
    assert np.all(np.diff(w.lm) >= 0)
    r = vio.synth.make_window(200, seed=5, ragged=True)
    counts = np.bincount(r.lm, minlength=200)
    assert counts.min() >= 1 and counts.max() <= 10 and len(set(counts)) > 3


def test_shard_window_partitions_landmarks(vio):
    w = vio.synth.make_window(101, seed=1, ragged=True)
    seen_obs, seen_lm = 0, 0
    for r in range(3):
        s = vio.synth.shard_window(w, r, 3)
        lo, hi = s.landmark_range
        assert s.n_landmarks == hi - lo and s.lm.max() < s.n_landmarks and s.lm.min() >= 0
        np.testing.assert_array_equal(s.inv_depth, w.inv_depth[lo:hi])
        np.testing.assert_array_equal(s.poses, w.poses)
        seen_obs += s.n_observations
        seen_lm += s.n_landmarks
    assert seen_obs == w.n_observations and seen_lm == 101


def test_algorithmic_bytes_formula(vio):
    # 4.1 MB at N = 20k, M = 80k
    b = vio.synth.algorithmic_bytes(20000, 80000)
    assert 4.0e6 < b < 4.2e6


def test_numpy_preintegration_matches_oracle_c(vio, oracle_lib):
    """The generator's numpy restatement of IntegrationBase (integration_base.h:54-158) against the oracle's C one."""
    rng = np.random.RandomState(0)
    n = 20
    acc = rng.normal(0, 1, (n + 1, 3)) + [0, 0, 9.8]
    gyr = rng.normal(0, 0.3, (n + 1, 3))
    ba, bg = rng.normal(0, 0.02, 3), rng.normal(0, 0.002, 3)
    dts = np.full(n, 0.005)
    py = vio.synth.preintegrate(acc[0], gyr[0], ba, bg, dts, acc[1:], gyr[1:])
    out = vio.VioPreint()
    f = oracle_lib.dll.vioo_preintegrate
    f.restype = None
    dp = lambda a: np.ascontiguousarray(a, dtype=np.float64).ctypes.data_as(C.POINTER(C.c_double))
    a0, g0, a1, g1 = (np.ascontiguousarray(x) for x in (acc[0], gyr[0], acc[1:], gyr[1:]))
    f(dp(a0), dp(g0), dp(ba), dp(bg), C.c_int(n), dp(dts), dp(a1), dp(g1), C.c_double(vio.synth.ACC_N),
      C.c_double(vio.synth.GYR_N), C.c_double(vio.synth.ACC_W), C.c_double(vio.synth.GYR_W), C.byref(out))
    assert abs(out.sum_dt - py["sum_dt"]) < 1e-15
    np.testing.assert_allclose(np.array(out.delta_p[:]), py["delta_p"], rtol=0, atol=1e-15)
    np.testing.assert_allclose(np.array(out.delta_q[:]), py["delta_q"], rtol=0, atol=1e-15)
    np.testing.assert_allclose(np.array(out.delta_v[:]), py["delta_v"], rtol=0, atol=1e-14)
    np.testing.assert_allclose(np.array(out.jacobian[:]).reshape(15, 15), py["jacobian"], rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(np.array(out.covariance[:]).reshape(15, 15), py["covariance"], rtol=1e-11, atol=1e-30)


def test_abi_preintegration_matches_numpy_and_oracle(vio, oracle_lib):
    """vio_preintegrate is host code inside the product library (as IntegrationBase is host code in the reference), so
    it can be exercised without a GPU: product C++ vs the generator's numpy vs the oracle's C."""
    lib = vio.load_hip()
    rng = np.random.RandomState(1)
    n = 20
    acc = rng.normal(0, 1, (n + 1, 3)) + [0, 0, 9.8]
    gyr = rng.normal(0, 0.3, (n + 1, 3))
    ba, bg = rng.normal(0, 0.02, 3), rng.normal(0, 0.002, 3)
    dts = np.full(n, 0.005)
    s = vio.synth
    got = lib.preintegrate(acc[0], gyr[0], ba, bg, dts, acc[1:], gyr[1:], s.ACC_N, s.GYR_N, s.ACC_W, s.GYR_W)
    py = s.preintegrate(acc[0], gyr[0], ba, bg, dts, acc[1:], gyr[1:])
    orc = oracle_lib.preintegrate(acc[0], gyr[0], ba, bg, dts, acc[1:], gyr[1:], s.ACC_N, s.GYR_N, s.ACC_W, s.GYR_W)
    for ref in (py, {k: np.array(getattr(orc, k)[:]) if k != "sum_dt" else orc.sum_dt for k in
                     ("sum_dt", "delta_p", "delta_q", "delta_v", "jacobian", "covariance")}):
        assert abs(got.sum_dt - ref["sum_dt"]) < 1e-15
        np.testing.assert_allclose(np.array(got.delta_p[:]), np.asarray(ref["delta_p"]).reshape(-1), rtol=0, atol=1e-15)
        np.testing.assert_allclose(np.array(got.delta_q[:]), np.asarray(ref["delta_q"]).reshape(-1), rtol=0, atol=1e-15)
        np.testing.assert_allclose(np.array(got.delta_v[:]), np.asarray(ref["delta_v"]).reshape(-1), rtol=0, atol=1e-14)
        np.testing.assert_allclose(np.array(got.jacobian[:]), np.asarray(ref["jacobian"]).reshape(-1), rtol=1e-12, atol=1e-15)
        np.testing.assert_allclose(np.array(got.covariance[:]), np.asarray(ref["covariance"]).reshape(-1), rtol=1e-11, atol=1e-30)


def test_observation_list_written_in_place(vio, oracle_lib):
    """vio_map_observations / vio_commit_observations (the library's own arrays filled by the caller's loop, one copy instead of two):
    the same window as through vio_set_observations, and the protocol's error paths — on the CPU restatement; the HIP library's
    arrays are pinned memory, its test is tests/test_gpu_parity.py::test_observation_list_written_in_place_on_the_gpu."""
    w = vio.synth.make_window(120, seed=8, ragged=True)
    a, b = oracle_lib.context(), oracle_lib.context()
    a.load(w)
    b.set_window(w.poses, w.speed_bias, w.ext)
    b.set_landmarks(w.inv_depth)
    lm, host, target, pi, pj = b.map_observations(w.n_observations)
    lm[:], host[:], target[:], pi[:], pj[:] = w.lm, w.host, w.target, w.pts_i, w.pts_j
    with pytest.raises(vio.VioError):
        b.linearize()                       # no list between map and commit
    b.commit_observations()
    for k, p in enumerate(w.preint):
        b.set_imu(k, p)
    b.set_prior(None)
    ra, rb = a.solve(10), b.solve(10)
    assert ra.final_chi2 == rb.final_chi2 and ra.iterations == rb.iterations
    assert np.array_equal(a.get_landmarks(), b.get_landmarks())
    with pytest.raises(vio.VioError):
        b.commit_observations()             # nothing mapped
    lm, host, target, pi, pj = b.map_observations(3)
    lm[:], host[:], target[:] = (0, 1, 500), (0, 0, 0), (1, 1, 1)
    with pytest.raises(vio.VioError):
        b.commit_observations()             # landmark index out of range


def test_mapping_protocol_on_the_oracle(vio, oracle_lib):
    tu.check_mapping_protocol(vio, oracle_lib, pytest)
