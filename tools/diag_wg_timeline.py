import ctypes as C, os, sys
import numpy as np
ROOT="/root/repo"
sys.path.insert(0, os.path.join(ROOT,"tests"))
from conftest import load_package
vio = load_package()
lib = vio.VioLib(os.path.join(ROOT,"visual-inertial-odometry_amd","csrc","diag","libvio_hip_stamps.so"),"vio_")
n=20000
g=int(os.environ.get("VIO_G_MAX","48"))
w = vio.synth.make_window(n, seed=42)
ctx = lib.context(); ctx.load(w)
for _ in range(3): ctx.linearize()
ctx.synchronize()
nb=(n+g-1)//g+10
buf=np.zeros((nb,16),dtype=np.uint64)
f=lib.dll.vio_debug_stamps; f.restype=C.c_int
assert f(ctx.h, buf.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_int64(nb))==0
st=buf.astype(np.int64)
v=st[:,5]>0
s0=st[v][:,0]; e=st[v][:,5]
# s_memtime is per XCD (unsynchronised): cluster the workgroups by clock domain, then look inside each
order=np.argsort(s0); s0=s0[order]; e=e[order]
cl=np.concatenate([[0],np.cumsum(np.diff(s0)>10_000_000)])
print("G",g,"blocks",int(v.sum()),"clock domains",int(cl.max()+1))
for k in range(int(cl.max())+1):
    m=cl==k
    a0=s0[m].min()
    so=np.sort(s0[m]-a0); eo=np.sort(e[m]-a0)
    print(" domain %d: %3d workgroups; starts %s ... ends %s; span %d" % (k, m.sum(), so[[0,len(so)//4,len(so)//2,3*len(so)//4,-1]].tolist(), eo[[0,len(eo)//2,-1]].tolist(), eo[-1]))
