"""Ad-hoc GPU exploration script (not a pytest test): HIP vs oracle on one window, step by step."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from conftest import load_package
vio = load_package()
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
orc = vio.VioLib(os.path.join(root, "oracle/liboracle.so"), "vioo_")
hip = vio.load_hip()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
ragged = len(sys.argv) > 2 and sys.argv[2] == "ragged"
extfix = 0 if (len(sys.argv) > 3 and sys.argv[3] == "free") else 1
w = vio.synth.make_window(N, seed=42, ragged=ragged)
print("N", N, "M", w.n_observations, "ragged", ragged, "ext_fixed", extfix, flush=True)
def rel(a, b):
    a = np.asarray(a); b = np.asarray(b)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)
co, ch = orc.context(ext_fixed=extfix), hip.context(ext_fixed=extfix)
co.load(w); ch.load(w)
co.linearize(); ch.linearize()
Ho, bo = co.get_schur_system(); Hh, bh = ch.get_schur_system()
print("Hs rel", rel(Hh, Ho), "bs rel", rel(bh, bo), "maxH", np.abs(Ho).max())
cam = [0,1,2,3,4,5] + [6+15*f+k for f in range(11) for k in range(6)]
print("Hs cam-block rel", rel(Hh[np.ix_(cam,cam)], Ho[np.ix_(cam,cam)]), "max", np.abs(Ho[np.ix_(cam,cam)]).max())
ho, blo = co.get_landmark_system(); hh, blh = ch.get_landmark_system()
print("hll rel", rel(hh, ho), "bl rel", rel(blh, blo))
go, do = co.get_pose_gradient(); gh, dh = ch.get_pose_gradient()
print("bpp rel", rel(gh, go), "diag rel", rel(dh, do))
print("init_lm", co.init_lm(), ch.init_lm())
chi, lam = co.init_lm()
co.solve_linear(lam); ch.solve_linear(lam)
dpo, dlo = co.get_delta(); dph, dlh = ch.get_delta()
print("dx pose abs", np.abs(dpo - dph).max(), "max", np.abs(dpo).max(), " dx lm abs", np.abs(dlo - dlh).max(), np.abs(dlo).max())
co.update_states(); ch.update_states()
po, so, eo = co.get_window(); ph, sh, eh = ch.get_window()
print("pose after update", np.abs(po - ph).max(), np.abs(so - sh).max(), np.abs(eo-eh).max(), np.abs(co.get_landmarks() - ch.get_landmarks()).max())
print("chi2", co.chi2(), ch.chi2())
print("eval", co.eval_step(), ch.eval_step())
co2, ch2 = orc.context(ext_fixed=extfix), hip.context(ext_fixed=extfix)
co2.load(w); ch2.load(w)
t=time.time(); ro = co2.solve(10); t1=time.time()-t
t=time.time(); rh = ch2.solve(10); t2=time.time()-t
print("solve time oracle %.4f hip %.4f (hip report %.3f ms, hessian %.3f ms)" % (t1, t2, rh.solve_ms, rh.hessian_ms))
print("oracle", ro.iterations, ro.trials, ro.accepted, ro.final_chi2, ro.final_lambda)
print("hip   ", rh.iterations, rh.trials, rh.accepted, rh.final_chi2, rh.final_lambda)
print(list(ro.chi2_trace[:ro.iterations])); print(list(rh.chi2_trace[:rh.iterations]))
po, so, eo = co2.get_window(); ph, sh, eh = ch2.get_window()
print("final pose diff", np.abs(po - ph).max(), np.abs(so - sh).max(), np.abs(co2.get_landmarks() - ch2.get_landmarks()).max())
t=time.time(); rh = ch2.solve(10); t2=time.time()-t
print("second solve hip %.4f s" % t2)
# marginalize
w2 = w.copy(); w2.poses, w2.speed_bias, w2.ext = po, so, eo; w2.inv_depth = co2.get_landmarks()
co2.load(w2); ch2.load(w2)
mo = co2.marginalize(vio.MARG_OLD); mh = ch2.marginalize(vio.MARG_OLD)
print("marg H rel", rel(mh["H"], mo["H"]), "b rel", rel(mh["b"], mo["b"]), "err norm", np.linalg.norm(mo["err"]), np.linalg.norm(mh["err"]))
# GN iteration timing
ch3 = hip.context(ext_fixed=extfix); ch3.load(w)
ch3.gn_iteration(lam); ch3.synchronize()
t=time.time()
for _ in range(20): ch3.gn_iteration(lam)
ch3.synchronize(); t2=(time.time()-t)/20
print("gn iteration %.1f us" % (t2*1e6))
