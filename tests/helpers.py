"""Shared helpers for the parity tests."""
import os

import numpy as np

from oracle import cnf_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GOLDEN_CASES = sorted(f[:-4] for f in os.listdir(GOLDEN) if f.endswith(".npz"))

# Parity bar (BASELINE.json north_star): 1e-4 relative in fp32.  Entries of a row that
# nearly cancel (the trace row sums n_in signed terms) carry fp32 noise proportional to the
# row's magnitude, not their own, so the error of an entry is measured against
# |ref| + rms(row):  |got - ref| <= RTOL * (|ref| + rms_row).
RTOL = 1e-4


def parity_err(got, ref):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    if ref.size == 0:
        return 0.0
    if ref.ndim == 1:
        scale = np.sqrt(np.mean(ref * ref))
    else:
        scale = np.sqrt(np.mean(ref * ref, axis=-1, keepdims=True))
    return float(np.max(np.abs(got - ref) / (np.abs(ref) + scale + 1e-30)))


def assert_parity(got, ref, what="", rtol=RTOL):
    assert np.all(np.isfinite(np.asarray(got, dtype=np.float64))), f"{what}: non-finite output"
    e = parity_err(got, ref)
    assert e <= rtol, f"{what}: parity error {e:.3e} > {rtol:g}"
    return e


def load_golden(name):
    g = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    net = O.Net(tuple(int(d) for d in g["dims"]), tuple(int(a) for a in g["acts"]))
    lam = g["lam"]
    cfg = O.Cfg(net, int(g["nvars"]), int(g["naugs"]), float(lam[0]), float(lam[1]), float(lam[2]),
                False, (float(g["tspan"][0]), float(g["tspan"][1])))
    return g, cfg


ACT_NAME = {v: k for k, v in O.ACT_NAMES.items()}


def make_icnf(cnf, cfg, *, jvp=False, kernel="auto", sol_kwargs=None, tag=None, rng=0):
    """Build the host-side ICNF that corresponds to an oracle Cfg."""
    layers = [cnf.Dense(i, o, ACT_NAME[a]) for i, o, a in zip(cfg.net.dims[:-1], cfg.net.dims[1:], cfg.net.acts)]
    nn = cnf.Chain(*layers)
    cm = cnf.HIPJacVecMatrixMode(kernel) if jvp else cnf.HIPVecJacMatrixMode(kernel)
    return cnf.construct(tag or cnf.FFJORD, nn, cfg.nvars, cfg.naugs, compute_mode=cm, tspan=cfg.tspan,
                         lambda1=cfg.lam1, lambda2=cfg.lam2, lambda3=cfg.lam3,
                         sol_kwargs=sol_kwargs or {}, rng=rng)
