import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
import continuousnf.jl_amd as cnf
from continuousnf.jl_amd import configs
cfg = configs.BASELINE[3]
B = int(sys.argv[2]) if len(sys.argv) > 2 else cfg.batch
rng = np.random.default_rng(5)
flat = configs.glorot_params(cfg.dims, 0)
kw = {}
if os.environ.get("PC_FIXED"):
    kw = dict(adaptive=False, dt=1.0 / int(os.environ["PC_FIXED"]))
elif os.environ.get("PC_DT"):
    kw = dict(dt=float(os.environ["PC_DT"]))
icnf = configs.build(cfg, kernel="mfma", sol_kwargs=kw)
dev = torch.device("cuda", 0)
xs = torch.tensor(rng.standard_normal((cfg.nvars, B)).astype(np.float32), device=dev)
eps = torch.tensor(rng.standard_normal((cfg.n_in, B)).astype(np.float32), device=dev)
out = cnf.inference(icnf, cnf.TrainMode(), xs, flat, {}, eps=eps, with_sums=True)
torch.cuda.synchronize()
logpx, regs, sums = out
t0 = time.perf_counter()
n = int(os.environ.get("PC_N", 50))
for _ in range(n):
    out = cnf.inference(icnf, cnf.TrainMode(), xs, flat, {}, eps=eps, with_sums=True)
    torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
st = icnf.last_stats if hasattr(icnf, "last_stats") else None
print("ms per solve %.4f" % (dt * 1e3), "stats", st, "sums", sums.cpu().numpy())
np.save(sys.argv[1], np.concatenate([logpx.cpu().numpy().ravel(), regs[0].cpu().numpy().ravel(), regs[1].cpu().numpy().ravel(), sums.cpu().numpy().ravel()]))
