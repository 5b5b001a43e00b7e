#!/usr/bin/env python3
"""More cases for tests/test_gpu_two_ranks.py (two real HIP ranks on one device, exchange through pinned host memory + gloo):
random sizes, ragged tracks, both kinds of landmark, with and without a (synthetic, rank-40) prior, other seeds.  Diagnostic: the
test's entry-wise bound on the sharded MargOldFrame (2e-5 of the largest entry) can trip on a tiny window with such a prior — the
marginalisation is ill-posed entry by entry — everything before it in the test is the protocol check.
  python tools/fuzz_two_ranks.py [cases] [seed]"""
import os
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_two_ranks as t  # noqa: E402
from conftest import ORACLE_DIR, load_package  # noqa: E402


def main():
    vio = load_package()
    hip = vio.load_hip()
    orc = vio.VioLib(os.path.join(ORACLE_DIR, "liboracle.so"), "vioo_")
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    rng = np.random.RandomState(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    bad = 0
    for case in range(cases):
        kind = "xyz" if rng.rand() < 0.35 else "invdepth"
        n = int(rng.choice([3, 40, 333, 1500, 7000]))
        ragged = bool(rng.randint(2))
        seed = int(rng.randint(1000))
        with_prior = bool(rng.randint(2))
        os.environ["VIO_TWO_RANK_SEED"] = str(seed)
        os.environ["VIO_TWO_RANK_PRIOR"] = "1" if with_prior else "0"
        try:
            with tempfile.TemporaryDirectory() as tmp:
                t.run_ranks_case(vio, hip, Path(tmp), kind, n, ragged)
            print("ok   case %d: %s n=%d ragged=%d seed=%d prior=%d" % (case, kind, n, ragged, seed, with_prior))
        except AssertionError as exc:
            bad += 1
            import traceback
            tb = traceback.extract_tb(exc.__traceback__)[-1]
            print("     (%s:%d: %s)%s" % (os.path.basename(tb.filename), tb.lineno, (tb.line or "")[:160], ("  " + str(exc)[:200]) if str(exc) else ""))
            print("FAIL case %d: %s n=%d ragged=%d seed=%d prior=%d: %s" % (case, kind, n, ragged, seed, with_prior, str(exc).splitlines()[0][:200] if str(exc) else "assertion"))
    print("failures:", bad)


if __name__ == "__main__":         # (the ranks are spawned: they import this file)
    main()
