import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import torch
import continuousnf.jl_amd as cnf
from continuousnf.jl_amd import configs
for i, B in ((3, 32), (3, 256), (5, 32)):
    wl = configs.BASELINE[i]
    flat = torch.from_numpy(configs.glorot_params(wl.dims, i, 0.05)).cuda()
    xs_h, eps_h = configs.synthetic_inputs(wl, B, i)
    xs = torch.from_numpy(xs_h).cuda()
    icnf = configs.build(wl, sol_kwargs=configs.README_TOLERANCES)
    fn = lambda: cnf.loss_and_grad(icnf, cnf.TestMode(), xs, flat, {})
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): fn()
    torch.cuda.synchronize()
    print(f"cfg{i} B={B}: TestMode loss_and_grad {(time.perf_counter()-t0)/3*1e3:.2f} ms, steps {icnf.last_stats['naccept']}", flush=True)
