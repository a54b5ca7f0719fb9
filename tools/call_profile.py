"""Where the host time of one small call goes: cProfile of loss / loss_and_grad on the reference's benchmark model
(benchmark/benchmarks.jl:24-59), 2000 calls each.   python tools/call_profile.py [train|test|grad|gradtest]"""
import cProfile, io, os, pstats, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import continuousnf.jl_amd as cnf
from continuousnf.jl_amd import layers

nn = cnf.Chain(cnf.Dense(16, 16, "tanh"))
icnf = cnf.construct(cnf.RNODE, nn, 8, 8, compute_mode=cnf.HIPVecJacMatrixMode(), tspan=(0.0, 13.0), steer_rate=0.1, lambda3=1e-2, rng=1)
ps, st = layers.setup(icnf.rng, nn, init="lux_v1")
dr = torch.from_numpy(np.random.default_rng(1).random((8, 64)).astype(np.float32)).cuda()
what = sys.argv[1] if len(sys.argv) > 1 else "test"
fn = {"train": lambda: cnf.loss(icnf, cnf.TrainMode(), dr, ps, st), "test": lambda: cnf.loss(icnf, cnf.TestMode(), dr, ps, st),
      "grad": lambda: cnf.loss_and_grad(icnf, cnf.TrainMode(), dr, ps, st), "gradtest": lambda: cnf.loss_and_grad(icnf, cnf.TestMode(), dr, ps, st)}[what]
for _ in range(50):
    fn()
pr = cProfile.Profile(); pr.enable()
for _ in range(2000):
    fn()
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:5000])
