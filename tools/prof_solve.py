"""Profiling driver for the one-launch solves: n adaptive solves (README tolerances) of a BASELINE configuration through
cnf_solve_tsit5, so that a rocprofv3 pass sees that configuration's solve kernel and nothing else of weight.  Prints one
JSON line with what a roofline needs besides the kernel's duration: evaluations per solve and the algorithmic work of one
evaluation (cnf_rhs_work).      python3 tools/prof_solve.py CFG [train|test|jvp] [n] [B]"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from continuousnf.jl_amd import _lib, configs

ci = int(sys.argv[1]) if len(sys.argv) > 1 else 5
mname = sys.argv[2] if len(sys.argv) > 2 else "train"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 16
wl = configs.BASELINE[ci]
B = int(sys.argv[4]) if len(sys.argv) > 4 else (8192 if ci == 4 else wl.batch)
train = mname != "test"
icnf = configs.build(wl, kernel="auto", jvp=(mname == "jvp"))
icnf.set_params(configs.glorot_params(wl.dims, ci))
l, h = _lib.lib(), icnf.handle()
m = 1 if train else 0
D = l.cnf_state_rows(h, m)
dev = torch.device("cuda", 0)
xs_h, eps_h = configs.synthetic_inputs(wl, B, ci)
u0 = torch.zeros(B, D, device=dev)
u0[:, :wl.nvars] = torch.from_numpy(np.ascontiguousarray(xs_h.T)).to(dev)
u0 = u0.reshape(-1)
eps = torch.from_numpy(np.ascontiguousarray(eps_h.T)).to(dev).reshape(-1)
out = torch.empty_like(u0)
sp = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
tol = configs.README_TOLERANCES
opts = _lib.cnf_solve_opts(wl.tspan[0], wl.tspan[1], tol["abstol"], tol["reltol"], 0.0, 1, 1 << 20, 0)
stats = _lib.cnf_solve_stats()
run = lambda: _lib.check(l.cnf_solve_tsit5(h, m, u0.data_ptr(), eps.data_ptr(), out.data_ptr(), B, C.byref(opts), C.byref(stats), sp), h)
run(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    run()
torch.cuda.synchronize()
el = (time.perf_counter() - t0) / n
fl, by = C.c_double(), C.c_double()
_lib.check(l.cnf_rhs_work(h, m, B, C.byref(fl), C.byref(by)))
print(json.dumps({"cfg": ci, "mode": mname, "B": B, "dims": list(wl.dims), "solves": n, "ms_per_solve": el * 1e3, "nf": stats.nf,
                  "naccept": stats.naccept, "nreject": stats.nreject, "launches": stats.launches,
                  "flops_per_rhs": fl.value, "bytes_per_rhs": by.value}), flush=True)
icnf.close()
