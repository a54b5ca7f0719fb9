"""The committed fixtures are reproduced by both oracle implementations (drift check), and
the float32 C restatement agrees with the float64 numpy oracle at the parity bar."""
import numpy as np
import pytest

from oracle import c_oracle as CO
from oracle import cnf_oracle as O
from tests.helpers import GOLDEN_CASES, assert_parity, load_golden


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_numpy_oracle_reproduces_golden(name):
    g, cfg = load_golden(name)
    flat, eps, u = (g[k].astype(np.float64) for k in ("flat", "eps", "u_train"))
    for jvp in (False, True):
        tag = "jvp" if jvp else "vjp"
        du = O.augmented_f_train(cfg.net, flat, u, eps, cfg.lam1 != 0, cfg.lam2 != 0, jvp)
        assert np.allclose(du, g[f"du_train_{tag}"], rtol=1e-12, atol=1e-13)
    du = O.augmented_f_test(cfg.net, flat, u[: cfg.n_in + 1])
    assert np.allclose(du, g["du_test"], rtol=1e-12, atol=1e-13)
    if cfg.n_in <= 32:
        fsol, logpx, regs, st = O.inference(cfg, flat, g["xs"].astype(np.float64), eps, True,
                                            dt=float(g["dt"]), adaptive=False)
        assert np.allclose(fsol, g["fsol_train_vjp"], rtol=1e-11, atol=1e-12)
        assert st.nf == int(g["nf_train_vjp"])


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_c_oracle_matches_golden(name):
    g, cfg = load_golden(name)
    for jvp in (False, True):
        cfg.use_jvp = jvp
        tag = "jvp" if jvp else "vjp"
        du = CO.rhs(cfg, g["flat"], g["u_train"], g["eps"], True)
        assert_parity(du, g[f"du_train_{tag}"], f"{name} du train {tag}", trace_row=cfg.n_in)
        u0 = O.inference_u0(cfg, g["xs"], True)
        fsol, st = CO.solve(cfg, g["flat"], u0, g["eps"], True, dt=float(g["dt"]), adaptive=False)
        assert st["nf"] == int(g[f"nf_train_{tag}"])
        assert_parity(fsol, g[f"fsol_train_{tag}"], f"{name} fsol train {tag}", trace_row=cfg.n_in)
        logpx, regs = CO.post(cfg, fsol, True)
        assert_parity(logpx, g[f"logpx_train_{tag}"], f"{name} logpx {tag}")
        assert_parity(regs, g[f"regs_train_{tag}"], f"{name} regs {tag}")
    cfg.use_jvp = False
    du = CO.rhs(cfg, g["flat"], g["u_train"][: cfg.n_in + 1], None, False)
    assert_parity(du, g["du_test"], f"{name} du test", trace_row=cfg.n_in)
    u0 = O.inference_u0(cfg, g["xs"], False)
    fsol, _ = CO.solve(cfg, g["flat"], u0, None, False, dt=float(g["dt"]), adaptive=False)
    assert_parity(fsol, g["fsol_test"], f"{name} fsol test", trace_row=cfg.n_in)


def test_c_oracle_adaptive_matches_float32_numpy_stepping():
    cfg, _, _ = O.baseline_cfg(2)
    rng = np.random.default_rng(11)
    flat = O.glorot_params(cfg.net, rng, np.float32, 0.1)
    xs = rng.standard_normal((cfg.nvars, 48)).astype(np.float32)
    eps = rng.standard_normal((cfg.n_in, 48)).astype(np.float32)
    u0 = O.inference_u0(cfg, xs, True)
    kw = dict(reltol=3.4526698e-4, abstol=1.1920929e-7)
    ref, st = O.tsit5_solve(cfg.rhs(flat, eps, True), u0, 0.0, 1.0, **kw)
    got, st2 = CO.solve(cfg, flat, u0, eps, True, **kw)
    assert abs(st.naccept - st2["naccept"]) <= 1 and st2["nf"] == 2 + 6 * (st2["naccept"] + st2["nreject"])
    ref64, _ = O.tsit5_solve(cfg.rhs(flat.astype(np.float64), eps.astype(np.float64), True),
                             u0.astype(np.float64), 0.0, 1.0, reltol=1e-10, abstol=1e-10)
    # at reltol 3.45e-4 the solver error itself bounds agreement
    assert_parity(got, ref64, "adaptive vs tight float64", rtol=5e-3, trace_row=cfg.n_in)
    assert_parity(got, ref, "adaptive C vs numpy float32", rtol=5e-3, trace_row=cfg.n_in)
