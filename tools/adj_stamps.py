"""In-kernel phase stamps of the MFMA pullback kernel (diagnostic build only):
    hipcc ... -DAM_STAMPS -c cnf_grad.hip ; link as build_abl/libcnf_STAMPS.so ;
    CNFHIP_LIB=$PWD/build_abl/libcnf_STAMPS.so python tools/adj_stamps.py [B]
Prints the s_memtime deltas between the barriers of workgroup 0 (ids in cnf_grad.hip: AM_STAMP)."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import continuousnf.jl_amd as cnf
from continuousnf.jl_amd import _lib
from oracle import cnf_oracle as O
from tests.helpers import make_icnf
cfg, _, _ = O.baseline_cfg(3)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
rng = np.random.default_rng(1)
flat = torch.from_numpy(O.glorot_params(cfg.net, rng, np.float32, 0.05)).cuda()
xs = torch.from_numpy(rng.standard_normal((cfg.nvars, B)).astype(np.float32)).cuda()
eps = torch.from_numpy(rng.standard_normal((cfg.n_in, B)).astype(np.float32)).cuda()
icnf = make_icnf(cnf, cfg, sol_kwargs=dict(adaptive=False, dt=0.25))
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
