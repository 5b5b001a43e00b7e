import os, sys
import numpy as np
sys.path.insert(0, "tests")
from conftest import load_package, ORACLE_DIR
import vio_testutil as tu
vio = load_package(); hip = vio.load_hip(); orc = vio.VioLib(os.path.join(ORACLE_DIR, "liboracle.so"), "vioo_")
for n, seed, ragged, ef in [(50, 11, False, 1), (300, 12, True, 1), (400, 13, False, 0), (2000, 14, False, 1), (150, 300, True, 1), (300, 301, True, 1)]:
    w = vio.synth.make_window(n, seed=seed, ragged=ragged)
    ch, co = hip.context(ext_fixed=ef), orc.context(ext_fixed=ef)
    ch.load(w); co.load(w)
    sh, rh = tu.run_solve(ch); so, ro = tu.run_solve(co)
    a, b = np.asarray(sh["lambda_trace"]), np.asarray(so["lambda_trace"])
    print(n, seed, "lambda rel dev per iteration:", ["%.1e" % x for x in np.abs(a - b) / np.abs(b)])
