"""Generates tests/golden/*.npz from the float64 oracle (oracle/cnf_oracle.py).

The reference cannot run in the build container (Julia absent) and holds no numeric
vectors, so these fixtures are outputs of the repo's own oracle, which tests/test_oracle.py
pins by known answers.  They freeze the oracle (drift check) and travel to the GPU box as
plain data: float32 inputs (params, u, eps, xs) and float64 expected outputs.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import cnf_oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
T = O.ACT_TANH

CASES = {
    # name: (dims, acts, nvars, naugs, lam1, lam2, lam3, tspan, B, fixed dt)
    "cfg1_readme": ((2, 6, 2), (T, T), 1, 1, 1e-2, 1e-2, 1e-2, (0.0, 13.0), 64, 13 / 128),
    "cfg2_regression": ((16, 48, 16), (T, T), 8, 8, 1e-2, 1e-2, 1e-2, (0.0, 1.0), 32, 1 / 32),
    "cfg3_headline_small": ((32, 128, 128, 32), (T, T, T), 32, 0, 1e-2, 1e-2, 0.0, (0.0, 1.0), 16, 1 / 16),
    "cfg4_ffjord_small": ((32, 128, 128, 32), (T, T, T), 32, 0, 0.0, 0.0, 0.0, (0.0, 1.0), 16, 1 / 16),
    "cfg5_exact_small": ((128, 384, 128), (T, T), 64, 64, 1e-2, 1e-2, 1e-2, (0.0, 1.0), 8, 1 / 8),
    "calltests_noaug": ((2, 2), (T,), 2, 0, 1e-2, 1e-2, 0.0, (0.0, 1.0), 4, 1 / 16),
    "calltests_aug": ((4, 4), (T,), 2, 2, 1e-2, 1e-2, 1e-2, (0.0, 1.0), 4, 1 / 16),
    "odd_mixed_acts": ((5, 7, 3, 5), (O.ACT_SOFTPLUS, O.ACT_SWISH, O.ACT_IDENTITY), 3, 2, 1e-2, 0.0, 1e-2,
                       (0.0, 0.5), 37, 1 / 32),
}


def make(name, spec, seed):
    dims, acts, nvars, naugs, l1, l2, l3, tspan, B, dt = spec
    rng = np.random.default_rng(seed)
    net = O.Net(dims, acts)
    flat32 = O.glorot_params(net, rng, np.float32, bias_scale=0.1)
    xs32 = rng.standard_normal((nvars, B)).astype(np.float32)
    eps32 = rng.standard_normal((nvars + naugs, B)).astype(np.float32)
    n_in = nvars + naugs
    utr32 = rng.standard_normal((n_in + 3, B)).astype(np.float32)
    flat, xs, eps, utr = (a.astype(np.float64) for a in (flat32, xs32, eps32, utr32))
    out = dict(dims=np.array(dims), acts=np.array(acts), nvars=nvars, naugs=naugs,
               lam=np.array([l1, l2, l3]), tspan=np.array(tspan), dt=dt,
               flat=flat32, xs=xs32, eps=eps32, u_train=utr32)
    for jvp in (False, True):
        cfg = O.Cfg(net, nvars, naugs, l1, l2, l3, jvp, tspan)
        tag = "jvp" if jvp else "vjp"
        out[f"du_train_{tag}"] = O.augmented_f_train(net, flat, utr, eps, l1 != 0, l2 != 0, jvp)
        fsol, logpx, (E, n, A), st = O.inference(cfg, flat, xs, eps, True, dt=dt, adaptive=False)
        out[f"fsol_train_{tag}"] = fsol
        out[f"logpx_train_{tag}"] = logpx
        out[f"regs_train_{tag}"] = np.stack([E, n, A])
        out[f"loss_train_{tag}"] = O.loss(cfg, logpx, (E, n, A), True)
        out[f"nf_train_{tag}"] = st.nf
    cfg = O.Cfg(net, nvars, naugs, l1, l2, l3, False, tspan)
    out["du_test"] = O.augmented_f_test(net, flat, utr[: n_in + 1])
    fsol, logpx, (_, _, A), st = O.inference(cfg, flat, xs, eps, False, dt=dt, adaptive=False)
    out["fsol_test"] = fsol
    out["logpx_test"] = logpx
    out["A_test"] = A
    out["loss_test"] = O.loss(cfg, logpx, (None, None, A), False)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    return path


if __name__ == "__main__":
    for i, (name, spec) in enumerate(CASES.items()):
        p = make(name, spec, 1000 + i)
        print(f"{name}: {os.path.getsize(p)} bytes")
