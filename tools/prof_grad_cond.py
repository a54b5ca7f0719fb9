"""loss_and_grad of a CONDITIONAL model of the headline width (CondRNODE, nn(vcat(z, ys)): 32 + 8 -> 128 -> 128 -> 32) at a few
batches.   python tools/prof_grad_cond.py [B ...]      (CNF_ADJ_GENERIC=1: the generic MFMA pullback instead of k_adj3)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import continuousnf.jl_amd as cnf
from continuousnf.jl_amd import configs

n_cond = 8
dims = (32 + n_cond, 128, 128, 32)
for B in [int(a) for a in sys.argv[1:]] or [32, 2048, 8192]:
    rng = np.random.default_rng(B)
    flat = torch.from_numpy(configs.glorot_params(dims, 3, 0.05)).cuda()
    xs = torch.from_numpy(rng.standard_normal((32, B)).astype(np.float32)).cuda()
    eps = torch.from_numpy(rng.standard_normal((32, B)).astype(np.float32)).cuda()
    ys = torch.from_numpy(rng.standard_normal((n_cond, B)).astype(np.float32)).cuda()
    nn = cnf.Chain(*[cnf.Dense(a, b, "tanh") for a, b in zip(dims[:-1], dims[1:])])
    icnf = cnf.construct(cnf.CondRNODE, nn, 32, 0, compute_mode=cnf.HIPVecJacMatrixMode(), tspan=(0.0, 1.0), lambda1=1e-2, lambda2=1e-2,
                         sol_kwargs=configs.README_TOLERANCES)
    out = {}
    for name, fn in (("loss", lambda: cnf.loss(icnf, cnf.TrainMode(), xs, ys, flat, {}, eps=eps)),
                     ("loss_and_grad", lambda: cnf.loss_and_grad(icnf, cnf.TrainMode(), xs, ys, flat, {}, eps=eps))):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            fn()
        torch.cuda.synchronize()
        out[name] = (time.perf_counter() - t0) / 4 * 1e3
    st = icnf.last_stats
    print(f"conditional 40-128-128-32 B={B}: loss {out['loss']:.2f} ms, loss_and_grad {out['loss_and_grad']:.2f} ms, "
          f"steps {st['naccept']}+{st['nreject']}", flush=True)
    icnf.close()
