"""A short, fixed-seed run of tools/fuzz_api_sequences.py: random sequences of C-ABI calls on one long-lived HIP context (partial
re-sets, getters, GN iterations, stepwise calls, both marginalisations, re-loads), every result compared with a fresh context
that is given the same inputs.  Guards the library's host-side bookkeeping: which mirror is newer, which plan is kept, the
marginalisation graph taken from the resident solve, inputs skipped because the context already holds them."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_random_call_sequences_agree_with_fresh_contexts(hip_lib):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_api_sequences.py"), "6", "30", "3"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "failures: 0" in out.stdout, out.stdout[-3000:]
    assert out.stdout.count("ok   sequence") == 6
