"""When and where the workgroups of one k_linearize launch ran (diagnostic build with in-kernel stamps):
s_memrealtime (100 MHz, device-wide) at the first and last stamp of every workgroup, XCC_ID and HW_ID."""
import ctypes as C, os, sys
import numpy as np
ROOT = "/root/repo"
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_package
vio = load_package()
lib = vio.VioLib(os.path.join(ROOT, "visual-inertial-odometry_amd", "csrc", "diag", "libvio_hip_stamps.so"), "vio_")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
w = vio.synth.make_window(n, seed=42)
ctx = lib.context(); ctx.load(w)
for _ in range(3): ctx.linearize()
if os.environ.get("VIO_DIAG_GN") == "1":        # the GN loop's k_linearize (carries the previous step's landmark update)
    _, lam = ctx.init_lm()
    for _ in range(4): ctx.gn_iteration(lam)
ctx.synchronize()
nb = (n + 95) // 96 + 10 if len(sys.argv) > 2 else 270
buf = np.zeros((nb, 16), dtype=np.uint64)
f = lib.dll.vio_debug_stamps; f.restype = C.c_int
assert f(ctx.h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_int64(nb)) == 0
st = buf.astype(np.int64)
v = (st[:, 9] > st[:, 8]) & (st[:, 8] > 0) & (st[:, 9] - st[:, 8] < 10_000_000)
idx = np.nonzero(v)[0]
t0, t1 = st[v][:, 8], st[v][:, 9]
xcc = st[v][:, 10] & 0xF
hw = st[v][:, 11]
cu, sh, se = (hw >> 8) & 0xF, (hw >> 12) & 0x1, (hw >> 13) & 0x7
cyc = st[v][:, 5] - st[v][:, 0]
base = t0.min()
print("workgroups stamped: %d (blocks %d..%d); kernel-internal span %.2f us; workgroup time median %.2f us (%d cycles)"
      % (v.sum(), idx.min(), idx.max(), (t1.max() - base) / 100.0, np.median(t1 - t0) / 100.0, np.median(cyc)))
print("start offsets (us): min %.2f  p25 %.2f  median %.2f  p75 %.2f  max %.2f" % tuple(np.percentile(t0 - base, [0, 25, 50, 75, 100]) / 100.0))
print("end offsets   (us): min %.2f  p25 %.2f  median %.2f  p75 %.2f  max %.2f" % tuple(np.percentile(t1 - base, [0, 25, 50, 75, 100]) / 100.0))
for x in range(8):
    m = xcc == x
    if m.any():
        slots = set(zip(se[m].tolist(), sh[m].tolist(), cu[m].tolist()))
        print(" XCC %d: %3d workgroups on %3d distinct (se,sh,cu); starts %.2f..%.2f us, ends %.2f..%.2f us"
              % (x, m.sum(), len(slots), (t0[m] - base).min() / 100.0, (t0[m] - base).max() / 100.0, (t1[m] - base).min() / 100.0, (t1[m] - base).max() / 100.0))
late = np.argsort(t0)[-12:]
print("latest starters: " + ", ".join("b%d@%.2f(xcc%d)" % (idx[i], (t0[i] - base) / 100.0, xcc[i]) for i in late))
