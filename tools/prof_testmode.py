"""Times the TestMode (exact trace) right-hand side and a full TestMode inference on config 3
(three layers: the MFMA exact-trace kernel of cnf_trace.hip).   python tools/prof_testmode.py [B]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
import continuousnf.jl_amd as cnf
from continuousnf.jl_amd import configs
wl = configs.BASELINE[3]
B = int(sys.argv[1]) if len(sys.argv) > 1 else wl.batch
rng = np.random.default_rng(1)
flat = configs.glorot_params(wl.dims, 1, 0.05)
u = torch.from_numpy(rng.standard_normal((wl.n_in + 1, B)).astype(np.float32)).cuda()
icnf = configs.build(wl)
f = lambda: cnf.augmented_f(u, flat, 0.0, icnf, cnf.TestMode(), icnf.nn, {}, None)
f(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5): f()
torch.cuda.synchronize()
print(f"cfg3 TestMode RHS B={B}: {(time.perf_counter()-t0)/5*1e3:.3f} ms")
# (column-major as the reference keeps it: a sample's rows contiguous -- no layout copy inside the call)
xs = torch.from_numpy(rng.standard_normal((B, wl.nvars)).astype(np.float32)).cuda().t()
ic = configs.build(wl, sol_kwargs=configs.README_TOLERANCES)
ps = torch.from_numpy(flat).cuda()
g = lambda: cnf.inference(ic, cnf.TestMode(), xs, ps, {})
g(); g(); torch.cuda.synchronize()
n = 5
t0 = time.perf_counter()
for _ in range(n): g()
torch.cuda.synchronize()
print(f"cfg3 TestMode inference B={B}: {(time.perf_counter()-t0)/n*1e3:.2f} ms", ic.last_stats)
