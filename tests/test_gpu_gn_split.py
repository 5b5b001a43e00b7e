"""The GN loop's split solve (off by default, VIO_GN_SPLIT=1): the speed-bias chain of an iteration's system — IMU factors,
prior and lambda only — eliminated by one more workgroup of k_linearize's grid, which forms the IMU items itself, and k_pose_solve_cs starting at
the camera block.  Same arithmetic, operation for operation: the states after four iterations equal the stepwise path's bit for bit, with and
without a prior; VIO_GN_SPLIT=2 (the chain in a launch of its own) checks the split of the solve kernel alone."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", ["1", "2"])
def test_split_solve_is_the_same_arithmetic(mode):
    env = dict(os.environ, VIO_GN_SPLIT=mode)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "diag_gn_split_compare.py")], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), r.stdout[-2000:] + r.stderr[-2000:]
