import os, sys, time, statistics
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package
vio = load_package(); hip = vio.load_hip()
ws = [vio.synth.make_window(20000, seed=42 + r, t0=1.0 + 0.1 * r) for r in range(12)]
c = hip.context()
a = {"map": [], "fill": [], "commit": [], "set_obs": []}
for r, w in enumerate(ws):
    c.set_window(w.poses, w.speed_bias, w.ext); c.set_landmarks(w.inv_depth)
    t0 = time.perf_counter()
    lm, host, target, pi, pj = c.map_observations(w.n_observations); t1 = time.perf_counter()
    lm[:], host[:], target[:], pi[:], pj[:] = w.lm, w.host, w.target, w.pts_i, w.pts_j; t2 = time.perf_counter()
    c.commit_observations(); t3 = time.perf_counter()
    w2 = ws[(r + 1) % len(ws)]
    c.set_landmarks(w2.inv_depth)
    t4 = time.perf_counter(); c.set_observations(w2.lm, w2.host, w2.target, w2.pts_i, w2.pts_j); t5 = time.perf_counter()
    if r:
        a["map"].append(t1 - t0); a["fill"].append(t2 - t1); a["commit"].append(t3 - t2); a["set_obs"].append(t5 - t4)
print({k: round(statistics.median(v) * 1e6) for k, v in a.items()})
