"""Times the exact-trace RHS kernel of config 3 (B = 8192) with HIP events around back-to-back C-ABI calls.
   python tools/time_trace.py [n]"""
import ctypes as C, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import continuousnf.jl_amd as cnf
from continuousnf.jl_amd import _lib, configs
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
wl = configs.BASELINE[3]
B = int(os.environ.get("PROF_B", wl.batch))
flat = configs.glorot_params(wl.dims, 1, 0.05)
icnf = configs.build(wl)
icnf.set_params(flat)
l, h = _lib.lib(), icnf.handle()
D = wl.n_in + 1
dev = torch.device("cuda", 0)
u = torch.randn(B * D, device=dev)
du = torch.empty_like(u)
sp = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3):
    _lib.check(l.cnf_rhs(h, 0, 0, u.data_ptr(), None, du.data_ptr(), B, sp), h)
e0.record()
for _ in range(n):
    _lib.check(l.cnf_rhs(h, 0, 0, u.data_ptr(), None, du.data_ptr(), B, sp), h)
e1.record(); e1.synchronize()
print(f"trace rhs: {e0.elapsed_time(e1) * 1e3 / n:.2f} us per launch")
