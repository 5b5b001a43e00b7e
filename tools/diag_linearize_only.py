import os, sys
ROOT="/root/repo"
sys.path.insert(0, os.path.join(ROOT,"tests"))
from conftest import load_package
vio = load_package()
hip = vio.load_hip()
w = vio.synth.make_window(20000, seed=42)
ctx = hip.context(); ctx.load(w)
for _ in range(5): ctx.linearize()
for kid,name in enumerate(hip.KERNELS[:3]):
    ctx.profile_begin(kid)
    for _ in range(50): ctx.linearize()
    ms,cnt = ctx.profile_end()
    print(name, round(ms/cnt*1e3,2), "us")
