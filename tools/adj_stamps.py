"""In-kernel phase stamps of the MFMA pullback kernel (diagnostic build only):
    hipcc ... -DAM_STAMPS -c cnf_grad.hip ; link as build_abl/libcnf_STAMPS.so ;
    CNFHIP_LIB=$PWD/build_abl/libcnf_STAMPS.so python tools/adj_stamps.py [B [cfg]]
Prints the s_memtime deltas between the barriers of workgroup 0 (ids in cnf_grad.hip: AM_STAMP)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import continuousnf.jl_amd as cnf
from continuousnf.jl_amd import _lib
from continuousnf.jl_amd import configs
wl = configs.BASELINE[int(sys.argv[2]) if len(sys.argv) > 2 else 3]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
flat = torch.from_numpy(configs.glorot_params(wl.dims, 1, 0.05)).cuda()
xs_h, eps_h = configs.synthetic_inputs(wl, B, 1)
xs, eps = torch.from_numpy(xs_h).cuda(), torch.from_numpy(eps_h).cuda()
icnf = configs.build(wl, sol_kwargs=dict(adaptive=False, dt=0.25))
for _ in range(2):
    cnf.loss_and_grad(icnf, cnf.TrainMode(), xs, flat, {}, eps=eps)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 64)()
l = _lib.lib()
l.cnf_debug_adj_stamps.restype = C.c_int
print("rc", l.cnf_debug_adj_stamps(buf, 64))
t = np.array(list(buf), dtype=np.int64)
n = 25
t = np.where(t == 0, np.nan, t.astype(float))
t = t[:n]
d = np.diff(t)
print("stamps", n, "total ticks", t[-1] - t[0])
print(" ".join(str(x) for x in d))
raw = np.array(list(buf), dtype=np.int64)
print("raw - min:", " ".join(str(int(x - raw[raw > 0].min())) if x > 0 else "-" for x in raw[:20]))
