"""The marginalisation's dense tail as the PRODUCT library runs it (csrc/host_dense.cpp compiled by hipcc, -ffp-contract=fast): on a host with
AVX-512 the rotations, tred2 and the two products keep their accumulators in 512-bit registers; the priors must be, byte for byte, what the
256-bit / baseline clones return (VIO_NO_AVX512=1) — the CPU tier checks the same routines under g++, where contraction is a compiler flag."""
import hashlib
import os
import subprocess
import sys

import pytest

from conftest import GOLDEN_DIR, ROOT

SNIPPET = r"""
import glob, hashlib, os, sys
import numpy as np
sys.path.insert(0, os.path.join(%(root)r, "tests"))
from conftest import load_package
import vio_testutil as tu
vio = load_package()
hip = vio.load_hip()
h = hashlib.sha256()
# the priors of a stream's windows (75 live rows in the steady state) and of the golden windows' marginalisation inputs
c = hip.context()
prior = None
for r in range(6):
    w = vio.synth.make_window(150, seed=300 + r, t0=1.0 + 0.1 * r, ragged=True)
    w.prior = prior
    c.load(w); c.solve(10)
    prior = c.marginalize(vio.MARG_OLD)
    for k in ("H", "b", "err", "jt_inv"):
        h.update(np.ascontiguousarray(prior[k]).tobytes())
print("DIGEST", h.hexdigest())
"""


@pytest.mark.gpu
def test_priors_do_not_depend_on_the_vector_width_of_the_host_tail():
    out = {}
    for name, env in (("wide", {}), ("narrow", {"VIO_NO_AVX512": "1"})):
        e = dict(os.environ, **env)
        e.pop("VIO_NO_AVX512", None) if name == "wide" else None
        r = subprocess.run([sys.executable, "-c", SNIPPET % {"root": ROOT}], capture_output=True, text=True, env=e, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        out[name] = [ln for ln in r.stdout.splitlines() if ln.startswith("DIGEST")][-1]
    assert out["wide"] == out["narrow"]
