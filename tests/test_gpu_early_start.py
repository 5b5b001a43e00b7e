"""k_pose_solve_c starts the chain's first level without a barrier in the GN loop and the stepwise solves (csrc/vio_pose_solve_chain.h: the chain waves
fetch their own tiles, the other waves copy the image and note its non-zero tiles on the way, one wave runs the previous step's test alone).  The
prologue it replaced — copy, barrier, lambda, barrier — is still there behind VIO_NO_EARLY_START=1 (gn_flags bit 5): both must leave the same bits."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def digest(env_extra):
    env = dict(os.environ)
    env.pop("VIO_NO_EARLY_START", None)
    env.update(env_extra)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "diag_early_start_compare.py")], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("digest ")]
    assert len(lines) == 1, p.stdout[-2000:]
    return lines[0]


def test_both_prologues_leave_the_same_bits():
    assert digest({}) == digest({"VIO_NO_EARLY_START": "1"})
