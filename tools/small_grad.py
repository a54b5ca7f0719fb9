import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import continuousnf.jl_amd as cnf
from continuousnf.jl_amd import configs
import dataclasses
wl = dataclasses.replace(configs.BASELINE[2], tspan=(0.0, 13.0), lambdas=(0.0, 0.0, 1e-2))
B = 32
flat = torch.from_numpy(configs.glorot_params(wl.dims, 2, 0.3)).cuda()
xs_h, eps_h = configs.synthetic_inputs(wl, B, 2)
xs, eps = torch.from_numpy(xs_h).cuda(), torch.from_numpy(eps_h).cuda()
icnf = configs.build(wl, sol_kwargs=configs.README_TOLERANCES)
for name, fn in (("loss", lambda: cnf.loss(icnf, cnf.TrainMode(), xs, flat, {}, eps=eps)),
                 ("loss_and_grad", lambda: cnf.loss_and_grad(icnf, cnf.TrainMode(), xs, flat, {}, eps=eps))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): fn()
    torch.cuda.synchronize()
    print(name, (time.perf_counter() - t0) / 20 * 1e3, "ms", icnf.last_stats)
