import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
import continuousnf.jl_amd as cnf
rng = np.random.default_rng(3)
dev = torch.device("cuda", 0)
worst = 0.0
for dims in ((32, 128, 128, 32), (20, 100, 100, 20), (30, 128, 64, 30), (5, 16, 128, 5)):
    n = dims[0]
    nn = cnf.Chain(*[cnf.Dense(a, b, "tanh") for a, b in zip(dims[:-1], dims[1:])])
    for tag, lam2 in ((cnf.RNODE, 1e-2), (cnf.FFJORD, 0.0)):
        for jvp in (False, True):
            for B in (1, 31, 33, 100, 1000, 8191):
                flat = (rng.standard_normal(sum(a * b + b for a, b in zip(dims[:-1], dims[1:]))) * 0.15).astype(np.float32)
                xs = torch.tensor(rng.standard_normal((n, B)).astype(np.float32), device=dev)
                eps = torch.tensor(rng.standard_normal((n, B)).astype(np.float32), device=dev)
                out = []
                for kernel in ("auto", "generic"):
                    cm = (cnf.HIPJacVecMatrixMode if jvp else cnf.HIPVecJacMatrixMode)(kernel)
                    kw = dict(lambda1=1e-2, lambda2=lam2, lambda3=0.0) if tag is cnf.RNODE else {}
                    ic = cnf.construct(tag, nn, n, 0, compute_mode=cm, sol_kwargs=dict(reltol=1e-4, abstol=1e-6), **kw)
                    lp, regs, sums = cnf.inference(ic, cnf.TrainMode(), xs, flat, {}, eps=eps, with_sums=True)
                    out.append((lp.cpu().numpy(), [r.cpu().numpy() for r in regs], sums.cpu().numpy(), ic.last_stats))
                    ic.close()
                a, b = out
                err = np.abs(a[0] - b[0]).max() / (1e-4 * np.abs(b[0]).max() + 1e-6)
                errs = np.abs(a[2] - b[2]).max() / (1e-4 * np.abs(b[2]).max() + 1e-6)
                worst = max(worst, err, errs)
                if err > 1.0 or errs > 1.0 or a[3]["launches"] > 3:
                    print("CHECK", dims, tag.__name__ if hasattr(tag, "__name__") else tag, jvp, B, err, errs, a[3], b[3]["naccept"])
print("worst err / bar", worst)
