#!/usr/bin/env python3
"""Diagnostic: Problem::Solve(10) in the reference's real regime (N = 150 / 300, ragged tracks, chained priors) and nothing else.

  python tools/diag_small_solve.py [n] [frames]          host wall clock of vio_solve per frame (median / min / max), iterations, trials
  rocprofv3 --kernel-trace -d <dir> -o s -- python3 tools/diag_small_solve.py 150 30
  python tools/diag_small_solve.py timeline <results.db> [solves]    the kernels of the last `solves` vio_solve calls: name, start offset, duration, gap
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def timeline(path, solves=1):
    import sqlite3
    from collections import defaultdict
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info('kernels')")]
    rows = db.execute("select name, start, end from kernels order by start").fetchall() if "start" in cols else None
    if rows is None:
        raise SystemExit("kernels view has no start/end: " + ",".join(cols))
    rows = [(n.split("(")[0], s, e) for n, s, e in rows]
    # a vio_solve begins with k_init_lm (the first linearisation is in front of it)
    starts = [i for i, r in enumerate(rows) if r[0] == "k_init_lm"]
    if not starts:
        raise SystemExit("no k_init_lm in the trace")
    agg_d, agg_g, cnt = defaultdict(float), defaultdict(float), defaultdict(int)
    spans = []
    for si in range(len(starts) - solves, len(starts)):
        a = starts[si]
        b = a + 1
        while b < len(rows) and rows[b][0] in ("k_pose_solve_c", "k_pose_solve_c2", "k_linearize", "k_linearize_g", "k_linearize_s", "k_linearize_gs", "k_reduce_c", "k_reduce", "k_assemble", "k_pose_solve"):
            b += 1
        seg = rows[a:b]
        spans.append((seg[-1][2] - seg[0][1]) / 1e3)
        prev = None
        for n, s, e in seg:
            agg_d[n] += e - s
            cnt[n] += 1
            if prev is not None:
                agg_g[n] += s - prev
            prev = e
        if si == len(starts) - 1:
            t0 = seg[0][1]
            prev = None
            for n, s, e in seg:
                print("%-18s start %8.2f us  dur %6.2f us  gap %5.2f us" % (n, (s - t0) / 1e3, (e - s) / 1e3, 0.0 if prev is None else (s - prev) / 1e3))
                prev = e
    print("--- over the last %d solves: device span k_init_lm .. last kernel: median %.1f us" % (solves, sorted(spans)[len(spans) // 2]))
    for n in sorted(agg_d, key=lambda k: -agg_d[k]):
        print("%-18s n/solve %5.1f  avg dur %6.2f us  avg gap before %5.2f us" % (n, cnt[n] / solves, agg_d[n] / cnt[n] / 1e3, agg_g[n] / cnt[n] / 1e3))


def main():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import load_package
    vio = load_package()
    hip = vio.load_hip()
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    frames = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    wins = [vio.synth.make_window(n, seed=300 + r, t0=1.0 + 0.1 * r, ragged=True) for r in range(22)]
    c = hip.context()
    prior = None
    ts, its, trs = [], [], []
    for r in range(frames + 3):
        w = wins[r % len(wins)]
        w.prior = prior
        c.load(w)
        c.linearize()
        c.synchronize()
        t = time.perf_counter()
        rep = c.solve(10)
        dt = time.perf_counter() - t
        prior = c.marginalize(vio.MARG_OLD)
        if r >= 3:
            ts.append(dt * 1e3)
            its.append(rep.iterations)
            trs.append(rep.trials)
    ts.sort()
    print("small_solve n=%d frames=%d  solve10_ms median %.4f min %.4f max %.4f  iterations %s trials %s items %s"
          % (n, frames, ts[len(ts) // 2], ts[0], ts[-1], sorted(set(its)), sorted(set(trs)), os.environ.get("VIO_SMALL_PLAN", "-")))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "timeline":
        timeline(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 1)
    else:
        main()
