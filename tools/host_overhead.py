"""Where the host time of one inference call goes (cProfile over 300 calls of the bench step)."""
import cProfile, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import continuousnf.jl_amd as cnf
from continuousnf.jl_amd import configs
wl = configs.BASELINE[3]
B = wl.batch
dev = torch.device("cuda", 0)
flat = configs.glorot_params(wl.dims, 12345)
xs_h, eps_h = configs.synthetic_inputs(wl, B, 1)
icnf = configs.build(wl, kernel="mfma", sol_kwargs=dict(configs.README_TOLERANCES))
xs = torch.from_numpy(np.ascontiguousarray(xs_h.T)).to(dev).t()
eps = torch.from_numpy(np.ascontiguousarray(eps_h.T)).to(dev).t()
ps = torch.from_numpy(flat).to(dev)
mode = cnf.TrainMode()
def step():
    return cnf.inference(icnf, mode, xs, ps, {}, eps=eps, with_sums=True)
for _ in range(20): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(300): step()
torch.cuda.synchronize()
print("us per call", (time.perf_counter() - t0) / 300 * 1e6)
pr = cProfile.Profile(); pr.enable()
for _ in range(300): step()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
