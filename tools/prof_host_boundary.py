"""The headline workload through the HOST-pointer entry points (cnf_inference_host: xs and eps cross PCIe in, logpx and the
regulariser rows out, every call) beside the device-pointer ones bench.py times.   python tools/prof_host_boundary.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import continuousnf.jl_amd as cnf
from continuousnf.jl_amd import configs

wl = configs.BASELINE[3]
B = 8192
flat = configs.glorot_params(wl.dims, 3, 0.05)
xs_h, eps_h = configs.synthetic_inputs(wl, B, 3)
icnf = configs.build(wl, sol_kwargs=configs.README_TOLERANCES)
flat_d = torch.from_numpy(flat).cuda()
xs_d, eps_d = torch.from_numpy(xs_h).cuda(), torch.from_numpy(eps_h).cuda()
for name, args in (("device pointers", (xs_d, flat_d, eps_d)), ("host pointers", (xs_h, flat, eps_h))):
    fn = lambda: cnf.inference(icnf, cnf.TrainMode(), args[0], args[1], {}, eps=args[2])
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    n = 50
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / n
    nf = icnf.last_stats["nf"]
    print(f"{name}: {el * 1e3:.3f} ms per inference (B = {B}, nf = {nf}) = {nf / el:.3e} RHS-evals/s", flush=True)
