import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
import continuousnf.jl_amd as cnf
from continuousnf.jl_amd import configs
wl = configs.BASELINE[3]
dev = torch.device("cuda", 0)
rng = np.random.default_rng(11)
icnfs = [configs.build(wl, kernel="mfma", sol_kwargs=dict(configs.README_TOLERANCES)),
         configs.build(wl, kernel="mfma", jvp=True, sol_kwargs=dict(configs.README_TOLERANCES))]
flat = torch.from_numpy(configs.glorot_params(wl.dims, 3)).to(dev)
t0 = time.time(); n = 0; launches = set()
while time.time() - t0 < 60:
    B = int(rng.choice([1, 7, 32, 33, 500, 4096, 8191, 8192, 8193, 12000]))
    xs = torch.randn(wl.nvars, B, device=dev); eps = torch.randn(wl.n_in, B, device=dev)
    ic = icnfs[n % 2]
    lp, regs, sums = cnf.inference(ic, cnf.TrainMode(), xs, flat, {}, eps=eps, with_sums=True)
    assert torch.isfinite(lp).all() and torch.isfinite(sums).all() and float(sums[4]) == B, (B, n)
    launches.add((B > 8192, ic.last_stats["launches"] <= 3))
    n += 1
print("soak:", n, "inferences in 60 s, all finite;", sorted(launches))
