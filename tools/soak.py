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
t0 = time.time(); tlast = t0; n = 0; launches = set(); nsub = 0; ngrad = 0
secs = float(os.environ.get("SOAK_S", "60"))
while time.time() - t0 < secs:
    B = int(rng.choice([1, 7, 32, 33, 500, 4096, 8191, 8192, 8193, 12000]))
    xs = torch.randn(wl.nvars, B, device=dev); eps = torch.randn(wl.n_in, B, device=dev)
    ic = icnfs[n % 2]
    lp, regs, sums = cnf.inference(ic, cnf.TrainMode(), xs, flat, {}, eps=eps, with_sums=True)
    assert torch.isfinite(lp).all() and torch.isfinite(sums).all() and float(sums[4]) == B, (B, n)
    launches.add((B > 8192, ic.last_stats["launches"] <= 3))
    if n % 5 == 0:
        # the same columns again, submitted among one to three others of random sizes and collected in turn: bit-equal
        depth = int(rng.integers(1, 4))
        batch = [(xs, eps, lp, sums)]
        for _ in range(depth - 1):
            Bj = int(rng.choice([1, 33, 1000, 8192, 8193]))
            xj = torch.randn(wl.nvars, Bj, device=dev); ej = torch.randn(wl.n_in, Bj, device=dev)
            batch.append((xj, ej, None, None))
        order = rng.permutation(len(batch))
        outs = [cnf.inference_submit(ic, cnf.TrainMode(), batch[i][0], flat, {}, eps=batch[i][1], with_sums=True) for i in order]
        for i, o in zip(order, outs):
            cnf.inference_collect(ic)
            torch.cuda.synchronize()
            assert torch.isfinite(o[0]).all() and float(o[2][4]) == batch[i][0].shape[1]
            if batch[i][2] is not None:
                assert torch.equal(o[0], batch[i][2]) and torch.equal(o[2], batch[i][3]), (B, n)
        nsub += len(batch)
    if n % 40 == 7 and B <= 8192:
        # the gradient of the same columns twice: bit-equal (no atomics anywhere on the gradient path), finite
        g = [cnf.loss_and_grad(icnfs[0], cnf.TrainMode(), xs, flat, {}, eps=eps) for _ in range(2)]
        assert g[0][0] == g[1][0] and torch.equal(g[0][1], g[1][1]) and torch.isfinite(g[0][1]).all(), (B, n)
        ngrad += 1
    n += 1
    if time.time() - tlast > 60:                      # (a line a minute: a silent run is taken to be hung on the GPU box)
        tlast = time.time(); print("  ...", n, "inferences,", int(tlast - t0), "s", flush=True)
print("soak:", n, "inferences +", nsub, "submitted +", ngrad, "gradient pairs in", int(secs), "s, all finite, submitted == synchronous, gradients bit-equal;", sorted(launches))
