"""Profiling driver: runs N plain RHS launches and/or N fused Tsit5 steps on BASELINE cfg3
(B = 8192) so that rocprofv3 sees only the kernels of interest.
usage: python3 tools/prof_rhs.py [rhs|step|adaptive|bench] [n] [kernel]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import continuousnf.jl_amd as cnf
from continuousnf.jl_amd import _lib
from continuousnf.jl_amd import configs

what = sys.argv[1] if len(sys.argv) > 1 else "step"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
kernel = sys.argv[3] if len(sys.argv) > 3 else "mfma"
cfgi = int(sys.argv[4]) if len(sys.argv) > 4 else 3
cfg = configs.BASELINE[cfgi]
B = int(os.environ.get("PROF_B", cfg.batch))
flat = configs.glorot_params(cfg.dims, 0)
ZERO = os.environ.get("PROF_ZERO") == "1"
if ZERO:
    flat = flat * 0
icnf = configs.build(cfg, kernel=kernel)
icnf.set_params(flat)
l, h = _lib.lib(), icnf.handle()
D = cfg.n_in + 3
dev = torch.device("cuda", 0)
u = torch.randn(B * D, device=dev)
eps = torch.randn(B * cfg.n_in, device=dev)
if ZERO:
    u.zero_(); eps.zero_()
du = torch.empty_like(u)
sp = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
k = {"mfma": 2, "generic": 1, "auto": 0}[kernel]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
if what == "rhs":
    for _ in range(3):
        _lib.check(l.cnf_rhs(h, 1, k, u.data_ptr(), eps.data_ptr(), du.data_ptr(), B, sp), h)
    e0.record()
    for _ in range(n):
        _lib.check(l.cnf_rhs(h, 1, k, u.data_ptr(), eps.data_ptr(), du.data_ptr(), B, sp), h)
    e1.record(); e1.synchronize()
    print(f"rhs: {e0.elapsed_time(e1) * 1e3 / n:.2f} us per launch")
elif what == "bench":
    # the launches bench.py times: adaptive solves of ITS inputs (seeds of SURVEY 8d-inputs) at the README tolerances
    wl = cfg
    flat_b = configs.glorot_params(wl.dims, 12345)
    xs_h, eps_h = configs.synthetic_inputs(wl, B, 1)
    icnf.set_params(flat_b)
    ub = torch.zeros(B * D, device=dev)
    ub.view(B, D)[:, :wl.nvars] = torch.from_numpy(np.ascontiguousarray(xs_h.T)).to(dev)
    eb = torch.from_numpy(np.ascontiguousarray(eps_h.T)).to(dev).contiguous()
    tol = configs.README_TOLERANCES
    opts = _lib.cnf_solve_opts(0.0, 1.0, tol["abstol"], tol["reltol"], 0.0, 1, 1 << 20, k)
    stats = _lib.cnf_solve_stats()
    run = lambda: _lib.check(l.cnf_solve_tsit5(h, 1, ub.data_ptr(), eb.data_ptr(), du.data_ptr(), B,
                                               C.byref(opts), C.byref(stats), sp), h)
    run(); torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        run()
    e1.record(); e1.synchronize()
    print(f"bench solve: {e0.elapsed_time(e1) * 1e3 / n:.1f} us per solve, nf={stats.nf} naccept={stats.naccept} "
          f"nreject={stats.nreject} launches={stats.launches}")
elif what == "adaptive":
    import time
    opts = _lib.cnf_solve_opts(0.0, 1.0, 1.1920929e-7, 3.4526698e-4, 0.0, 1, 1 << 20, k)
    stats = _lib.cnf_solve_stats()
    u.view(B, D)[:, cfg.n_in:] = 0
    run = lambda: _lib.check(l.cnf_solve_tsit5(h, 1, u.data_ptr(), eps.data_ptr(), du.data_ptr(), B,
                                               C.byref(opts), C.byref(stats), sp), h)
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        run()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / n
    print(f"adaptive: {el * 1e6:.1f} us per solve, nf={stats.nf} naccept={stats.naccept} nreject={stats.nreject} "
          f"launches={stats.launches} -> {stats.nf / el:.0f} RHS-evals/s")
else:
    opts = _lib.cnf_solve_opts(0.0, 1.0, 0.0, 0.0, 1.0 / n, 0, 1 << 20, k)
    stats = _lib.cnf_solve_stats()
    run = lambda: _lib.check(l.cnf_solve_tsit5(h, 1, u.data_ptr(), eps.data_ptr(), du.data_ptr(), B,
                                               C.byref(opts), C.byref(stats), sp), h)
    run()
    e0.record(); run(); e1.record(); e1.synchronize()
    print(f"step: {e0.elapsed_time(e1) * 1e3 / n:.2f} us per step, nf={stats.nf}")
