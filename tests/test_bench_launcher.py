"""bench.py --gpus N without a launcher (the driver's own command): the parent polls its rank processes and ends all of them on the
first failure — a rank that dies must not leave the others (and the parent) inside a collective until the backend's timeout."""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_a_failing_rank_ends_the_run_within_seconds():
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None); env.pop("RANK", None)
    env["VIO_BENCH_FAIL_RANK"] = "2"            # ranks 0, 1, 3 "hang"; rank 2 exits with code 3
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "5", "--warmup", "1"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    dt = time.time() - t0
    assert p.returncode == 3, (p.returncode, p.stderr[-500:])
    assert dt < 30.0, dt
