"""Diagnostic: which HIP runtime(s) end up in the process, by import order."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys
sys.path.insert(0, os.path.join(%r, "tests"))
from conftest import load_package
vio = load_package()
order = sys.argv[1]
def maps():
    return sorted({l.split()[-1] for l in open("/proc/self/maps") if "amdhip64" in l or "librccl" in l or "libhsa-runtime" in l})
if order == "lib_first":
    lib = vio.load_hip()
    import torch
else:
    import torch
    lib = vio.load_hip()
print(order, maps())
try:
    c = lib.context(); print(" create before torch.cuda init: ok"); c.close()
except Exception as e: print(" create before torch.cuda init:", e)
print(" torch.cuda.is_available:", torch.cuda.is_available(), torch.version.hip)
x = torch.zeros(4, device="cuda"); torch.cuda.synchronize()
try:
    c = lib.context(); print(" create after torch.cuda init: ok"); c.close()
except Exception as e: print(" create after torch.cuda init:", e)
if len(sys.argv) > 2:
    import torch.distributed as dist
    os.environ["MASTER_ADDR"]="127.0.0.1"; os.environ["MASTER_PORT"]="29533"
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda",0))
    t = torch.ones(4, device="cuda", dtype=torch.float64); dist.all_reduce(t); torch.cuda.synchronize()
    try:
        c = lib.context(); print(" create after rccl init: ok"); c.close()
    except Exception as e: print(" create after rccl init:", e)
    dist.destroy_process_group()
print(maps())
''' % ROOT
for args in (["lib_first"], ["torch_first"], ["lib_first", "rccl"], ["torch_first", "rccl"]):
    r = subprocess.run([sys.executable, "-c", code] + args, capture_output=True, text=True)
    print(r.stdout.strip()); print(r.stderr.strip()[-300:] if r.returncode else "")
