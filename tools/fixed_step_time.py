import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
import continuousnf.jl_amd as cnf
from continuousnf.jl_amd import _lib, configs
cfg = configs.BASELINE[3]; B = cfg.batch
icnf = configs.build(cfg, kernel="mfma", jvp=os.environ.get("FST_JVP") == "1"); icnf.set_params(configs.glorot_params(cfg.dims, 0))
l, h = _lib.lib(), icnf.handle()
D = cfg.n_in + 3; dev = torch.device("cuda", 0)
u = torch.randn(B * D, device=dev) * 0.5; u.view(B, D)[:, cfg.n_in:] = 0
eps = torch.randn(B * cfg.n_in, device=dev); du = torch.empty_like(u)
sp = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
for n in (32, 128):
    opts = _lib.cnf_solve_opts(0.0, 1.0, 0.0, 0.0, 1.0 / n, 0, 1 << 20, 2)
    stats = _lib.cnf_solve_stats()
    run = lambda: _lib.check(l.cnf_solve_tsit5(h, 1, u.data_ptr(), eps.data_ptr(), du.data_ptr(), B, C.byref(opts), C.byref(stats), sp), h)
    run(); run()
    _lib.check(l.cnf_solve_kernel_time(h, 1, None, None), h)
    for _ in range(5): run()
    us, k = C.c_float(), C.c_int()
    _lib.check(l.cnf_solve_kernel_time(h, 0, C.byref(us), C.byref(k)), h)
    print(n, "steps: kernel", us.value, "us ->", us.value / n, "us per step", "launches", stats.launches)
