"""Soak of the wave-local kernels (k_solve_wave and its GRAD forms): random batch sizes across the one-workgroup / one-wave-
per-workgroup boundary, random spans, every mode; each call twice -- results bit-equal, finite, one or two launches, no fallback.
    SOAK_S=120 python tools/soak_small.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import continuousnf.jl_amd as cnf

dev = torch.device("cuda", 0)
rng = np.random.default_rng(5)
nets = [((16, 48, 16), 8, 8), ((2, 6, 2), 1, 1), ((16, 16), 8, 8), ((7, 20, 7), 7, 0)]
models = []
for dims, nvars, naugs in nets:
    layers = [cnf.Dense(i, o, "tanh") for i, o in zip(dims[:-1], dims[1:])]
    for jvp in (False, True):
        cm = cnf.HIPJacVecMatrixMode() if jvp else cnf.HIPVecJacMatrixMode()
        ic = cnf.construct(cnf.RNODE, cnf.Chain(*layers), nvars, naugs, compute_mode=cm, tspan=(0.0, 13.0), lambda3=1e-2 if naugs else 0.0, rng=3)
        ps = (0.6 * cnf.setup(int(rng.integers(1 << 30)), cnf.Chain(*layers))[0]).astype(np.float32)
        models.append((ic, ps, nvars, nvars + naugs))
t0 = time.time(); tlast = t0; n = 0; ng = 0
secs = float(os.environ.get("SOAK_S", "60"))
while time.time() - t0 < secs:
    ic, ps, nvars, n_in = models[n % len(models)]
    B = int(rng.choice([1, 15, 16, 17, 32, 33, 48, 64, 65, 100, 1000, 2048, 4100]))
    ic.tspan = (0.0, float(rng.uniform(0.5, 13.0)))
    xs = torch.randn(nvars, B, device=dev); eps = torch.randn(n_in, B, device=dev)
    for mode in (cnf.TrainMode(), cnf.TestMode()):
        kw = dict(eps=eps) if isinstance(mode, cnf.TrainMode) else {}
        a = cnf.inference(ic, mode, xs, ps, {}, **kw)[0].clone()
        assert ic.last_stats["launches"] == 1, (B, ic.last_stats)
        b = cnf.inference(ic, mode, xs, ps, {}, **kw)[0]
        assert torch.isfinite(a).all() and torch.equal(a, b), (B, n)
        v1, g1 = cnf.loss_and_grad(ic, mode, xs, ps, {}, **kw)
        assert ic.last_stats["launches"] <= 2, (B, ic.last_stats)
        v2, g2 = cnf.loss_and_grad(ic, mode, xs, ps, {}, **kw)
        assert np.isfinite(v1) and v1 == v2 and torch.isfinite(g1).all() and torch.equal(g1, g2), (B, n)
        ng += 2
    assert ic.solve_fallbacks() == 0
    n += 1
    if time.time() - tlast > 30:
        tlast = time.time(); print(f"{tlast - t0:.0f} s: {n} rounds, {ng} gradients", flush=True)
print(f"soak ok: {n} rounds, {4 * n} inferences, {ng} gradients in {time.time() - t0:.0f} s")
