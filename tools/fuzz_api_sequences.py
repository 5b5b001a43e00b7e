#!/usr/bin/env python3
"""Randomised sequences of C-ABI calls on one HIP context (diagnostic; the fixed cases live in tests/).  What it is after is
the host-side state machine of the library — which inputs are newer on the host, which results are newer on the device,
plans kept or rebuilt, the marginalisation graph taken from the resident solve, inputs that are bitwise the ones already held
— under call orders no test spells out: partial re-sets (only the window, only the landmarks, only the prior, one IMU factor),
getters in between, GN iterations, stepwise calls, MargOldFrame / MargNewFrame at any point, the same window loaded twice, a
different window.

The check: whenever the long-lived context is asked for a result (solve, marginalise, chi2, a stepwise LM step), a FRESH context
is given exactly what the long-lived one should hold at that point — the inputs the sequence set last, the states / landmarks /
prior vectors the long-lived context returns — and asked the same thing.  Same library, same arithmetic, same inputs: the two
must agree to rounding (bit for bit on most paths), however ill-conditioned the window is.  The oracle runs the same sequence
beside them: an error must be an error in both.

  python tools/fuzz_api_sequences.py [sequences] [steps] [seed] [only this sequence, verbose]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import ORACLE_DIR, load_package  # noqa: E402

vio = load_package()
hip = vio.load_hip()
orc = vio.VioLib(os.path.join(ORACLE_DIR, "liboracle.so"), "vioo_")
n_seq = int(sys.argv[1]) if len(sys.argv) > 1 else 10
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
seed0 = int(sys.argv[3]) if len(sys.argv) > 3 else 1
only = int(sys.argv[4]) if len(sys.argv) > 4 else -1


def prior_of(seed):
    w = vio.synth.make_window(200, seed=seed, t0=0.9)
    c = orc.context()
    c.load(w)
    c.solve(10)
    return c.marginalize(vio.MARG_OLD)


PRIORS = [prior_of(71), prior_of(72)]
TIGHT = 1e-9        # long-lived context against a fresh one, relative to the result's scale


def rel(a, b):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)) if a.size else 0.0


bad = 0
for seq in range(n_seq):
    if only >= 0 and seq != only:
        continue
    rng = np.random.RandomState(1000 * seed0 + seq)
    kw = dict(ext_fixed=int(rng.randint(2)), loss_type=int(rng.choice([0, 2])))
    ch, co = hip.context(**kw), orc.context(**kw)
    xyz = rng.rand() < 0.3            # a sequence of XYZ-landmark windows (no MargOldFrame for those)
    mk = (lambda n, **k: vio.synth.make_window_xyz(n, obs_per_landmark=4, **k)) if xyz else vio.synth.make_window
    get_lm = (lambda c: c.get_landmarks_xyz()) if xyz else (lambda c: c.get_landmarks())
    set_lm = (lambda c, v: c.set_landmarks_xyz(v)) if xyz else (lambda c, v: c.set_landmarks(v))
    windows = [mk(int(rng.choice([60, 150, 400, 1200])), seed=300 + 10 * seq + q, ragged=bool(rng.randint(2))) for q in range(2)]
    for w in windows:
        w.prior = PRIORS[rng.randint(2)] if rng.rand() < 0.5 else None
    # the model of what the contexts were given last
    model = dict(w=windows[0], preint=list(windows[0].preint), prior=windows[0].prior)
    for c in (ch, co):
        c.load(model["w"])
    log, state = [], dict(ok=True)

    def fresh_twin():
        """A new HIP context holding what the long-lived one should hold now."""
        f = hip.context(**kw)
        w = model["w"].copy()
        w.poses, w.speed_bias, w.ext = ch.get_window()
        if xyz:
            w.xyz = get_lm(ch)
        else:
            w.inv_depth = get_lm(ch)
        w.preint = list(model["preint"])
        if model["prior"] is not None:
            p = dict(model["prior"])
            b, e = ch.get_prior()
            p["b"], p["err"] = b[:156].copy(), e.copy()
            w.prior = p
        else:
            w.prior = None
        f.load(w)
        return f

    def fail(msg):
        state["ok"] = False
        print("  MISMATCH after %s: %s" % (" > ".join(log[-8:]), msg))

    def run(name, fn):
        """the operation on the long-lived HIP context and on the oracle; errors must be errors on both"""
        log.append(name)
        out = []
        for c in (ch, co):
            try:
                out.append((fn(c), None))
            except vio.VioError as exc:
                out.append((None, exc))
        if (out[0][1] is None) != (out[1][1] is None):
            if "NOT_FINITE" in str(out[0][1] or out[1][1]):
                # a window at the edge of numerical breakdown (a landmark running off): which side of it an implementation
                # lands on is not a property of the host bookkeeping; the sequence ends here
                print("  (sequence ends: %s non-finite on one side only after %s)" % (name, " > ".join(log[-4:])))
                state["stop"] = True
                return None, None
            fail("HIP %s, oracle %s" % (out[0][1] or "ok", out[1][1] or "ok"))
            return None, None
        return out[0][0], out[1][0]

    def note(name, a, b, what):
        if only >= 0:
            print("    %-16s %-10s %.2e" % (name, what, rel(a, b)))

    for step in range(n_steps):
        if not state["ok"] or state.get("stop"):
            break
        op = rng.choice(["solve", "gn", "get", "chi2", "set_window", "set_landmarks", "set_prior", "reload_same", "load_other", "marg_old", "marg_new",
                         "stepwise", "set_imu", "set_config", "linearize"], p=[.15, .1, .08, .08, .08, .07, .07, .05, .06, .08, .05, .05, .03, .03, .02])
        if op == "linearize":
            # (round 6: a vio_solve that follows a vio_linearize on the same state and graph starts from that system — and must not when anything
            #  came in between: whatever follows in this sequence is compared with a fresh context as always)
            run("linearize", lambda c: c.linearize())
            continue
        if op in ("solve", "marg_old", "marg_new", "chi2", "stepwise"):
            if (op == "marg_new" and model["prior"] is None) or (op == "marg_old" and xyz):
                continue
            twin = fresh_twin()          # (its getters settle whatever the long-lived context still owed)
            if op == "solve":
                its = int(rng.randint(1, 6))
                f = lambda c: (lambda r: (r.iterations, r.trials, r.final_chi2) + c.get_window()[:2] + (get_lm(c),))(c.solve(its))   # noqa: E731
                name = "solve(%d)" % its
            elif op == "chi2":
                f, name = (lambda c: (c.chi2(),)), "chi2"
            elif op == "stepwise":
                def f(c):
                    c.linearize()
                    chi, lam = c.init_lm()
                    c.solve_linear(lam)
                    c.update_states()
                    okk, chi2, lam2 = c.eval_step()
                    if not okk:
                        c.rollback_states()
                    return (chi, lam, float(okk), chi2) + c.get_window()[:2]
                name = "stepwise"
            else:
                kind = vio.MARG_OLD if op == "marg_old" else vio.MARG_SECOND_NEW
                f, name = (lambda c: (lambda m: (m["H"], m["b"], m["err"]))(c.marginalize(kind))), op
            a, b = run(name, f)
            t = None
            try:
                t = f(twin)
            except vio.VioError as exc:
                if a is not None:
                    fail("the fresh context raised %s" % exc)
            if a is not None and t is not None:
                for q, (x, y) in enumerate(zip(a, t)):
                    note(name, x, y, "twin[%d]" % q)
                    if rel(x, y) > TIGHT:
                        fail("%s: long-lived context and fresh context differ in output %d by %.3e (relative)" % (name, q, rel(x, y)))
                        break
            del twin
        elif op == "gn":
            k = int(rng.randint(1, 4))
            lam = float(rng.choice([1e5, 5e5]))
            run("gn x%d" % k, lambda c: [c.gn_iteration(lam) for _ in range(k)] and None)
        elif op == "get":
            a, b = run("get", lambda c: c.get_window()[:2] + (get_lm(c),))
            if a is not None:
                for x, y in zip(a, b):
                    note("get", x, y, "oracle")
        elif op == "set_window":
            p, s, e = ch.get_window()
            p2 = p.copy()
            p2[:, 0:3] += rng.normal(0, 2e-3, size=(11, 3))
            run("set_window", lambda c: c.set_window(p2, s, e))
        elif op == "set_landmarks":
            lm0 = get_lm(ch)
            lm = lm0 * (1.0 + 1e-3 * rng.normal(size=lm0.shape)) if rng.rand() < 0.7 else lm0
            run("set_landmarks", lambda c: set_lm(c, lm))
        elif op == "set_prior":
            pr = PRIORS[rng.randint(2)] if rng.rand() < 0.7 else None
            model["prior"] = pr
            run("set_prior(%s)" % ("yes" if pr is not None else "none"), lambda c: c.set_prior(pr))
        elif op == "set_imu":
            k = int(rng.randint(10))
            pre = model["w"].preint[k] if rng.rand() < 0.7 else None
            model["preint"][k] = pre
            run("set_imu(%d,%s)" % (k, "yes" if pre is not None else "none"), lambda c: c.set_imu(k, pre))
        elif op == "set_config":          # vio_set_config on the living context: other loss, extrinsic fixed or free
            kw = dict(ext_fixed=int(rng.randint(2)), loss_type=int(rng.choice([0, 1, 2])))
            run("set_config(%d,%d)" % (kw["ext_fixed"], kw["loss_type"]), lambda c: c.set_config(**kw))
        elif op in ("reload_same", "load_other"):
            if op == "load_other":
                model["w"] = windows[1] if model["w"] is windows[0] else windows[0]
            model["preint"], model["prior"] = list(model["w"].preint), model["w"].prior
            run("load(%s)" % ("same" if op == "reload_same" else "other"), lambda c: c.load(model["w"]))
    bad += 0 if state["ok"] else 1
    print("%s sequence %2d: %s%s, %d steps: %s" % ("ok  " if state["ok"] else "FAIL", seq, "xyz " if xyz else "", kw, len(log), " ".join(log[:14]) + (" ..." if len(log) > 14 else "")))
print("failures:", bad)
