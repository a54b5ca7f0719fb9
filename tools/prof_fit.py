"""Time per gradient step of `mlj.fit` (src/exts/mlj_ext/core_icnf.jl:14-73: batch 32, Lion) on a few networks.
    python tools/prof_fit.py [epochs]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import continuousnf.jl_amd as cnf
from continuousnf.jl_amd import mlj

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 3
rng = np.random.default_rng(0)
for name, nvars, naugs, dims, tspan in (("regression 8 + 8, 16-48-16, tspan (0, 13)", 8, 8, (16, 48, 16), (0.0, 13.0)),
                                        ("headline network 32-128-128-32", 32, 0, (32, 128, 128, 32), (0.0, 1.0)),
                                        ("config 5's network 128-384-128", 64, 64, (128, 384, 128), (0.0, 1.0))):
    n = 1024
    r = rng.beta(2.0, 4.0, size=(nvars, n)).astype(np.float32)
    nn = cnf.Chain(*[cnf.Dense(a, b, "tanh") for a, b in zip(dims[:-1], dims[1:])])
    icnf = cnf.construct(cnf.RNODE, nn, nvars, naugs, compute_mode=cnf.HIPVecJacMatrixMode(), tspan=tspan, rng=1,
                         sol_kwargs=dict(reltol=float(np.sqrt(np.finfo(np.float32).eps)), abstol=float(np.finfo(np.float32).eps)))
    model = mlj.ICNFModel(icnf, n_epochs=1)
    mlj.fit(model, 0, r.T)                       # warm-up: allocations, first launches
    model = mlj.ICNFModel(icnf, n_epochs=epochs)
    t0 = time.perf_counter()
    _, _, report = mlj.fit(model, 0, r.T)
    el = time.perf_counter() - t0
    it = report["stats"]["iterations"]
    print(f"{name}: {it} gradient steps (batch {model.batch_size}) in {el:.2f} s = {el / it * 1e3:.2f} ms per step, "
          f"pipelined = {report['stats']['pipelined']}, last loss {float(report['losses'][-1]):.3f}", flush=True)
    icnf.close()
