"""Known-answer checks that pin the CPU oracle.  The reference ships no
numeric vectors for this path and Julia cannot run here, so the oracle is checked against
independent mathematics: autograd, finite differences, the linear closed form, scipy."""
import math

import numpy as np
import pytest
import scipy.integrate
import scipy.linalg
import scipy.stats
import torch

from oracle import cnf_oracle as O


def _net(dims, act=O.ACT_TANH):
    return O.Net(tuple(dims), (act,) * (len(dims) - 1))


def _torch_mlp(net, flat):
    Ws, bs = O.unflatten_params(net, flat)
    Ws = [torch.tensor(np.array(W)) for W in Ws]
    bs = [torch.tensor(np.array(b)) for b in bs]
    fns = {O.ACT_IDENTITY: lambda a: a, O.ACT_TANH: torch.tanh, O.ACT_SIGMOID: torch.sigmoid,
           O.ACT_SOFTPLUS: torch.nn.functional.softplus, O.ACT_RELU: torch.relu,
           O.ACT_SWISH: torch.nn.functional.silu, O.ACT_ELU: torch.nn.functional.elu}

    def f(z):  # z: (n_in,)
        h = z
        for W, b, k in zip(Ws, bs, net.acts):
            h = fns[k](W @ h + b)
        return h
    return f


def test_tsit5_tableau_order_conditions():
    c, b = np.array(O.TSIT5_C), np.array(O.TSIT5_B)
    A = np.zeros((7, 7))
    for i, row in enumerate(O.TSIT5_A):
        A[i, :len(row)] = row
    assert np.allclose(A.sum(1), c, atol=1e-15)
    assert abs(b.sum() - 1) < 1e-15
    assert abs(sum(O.TSIT5_BTILDE)) < 1e-15
    for k in range(1, 5):
        assert abs(b @ c**k - 1 / (k + 1)) < 1e-14
    assert abs(b @ A @ c - 1 / 6) < 1e-14
    assert abs(b @ A @ c**2 - 1 / 12) < 1e-14
    assert abs(b @ A @ A @ c - 1 / 24) < 1e-14
    assert abs(b @ A @ A @ A @ c - 1 / 120) < 1e-14
    # the embedded solution b - btilde is 4th order: order-5 condition fails by ~6e-4
    bh = b - np.array(O.TSIT5_BTILDE)
    for k in range(1, 4):
        assert abs(bh @ c**k - 1 / (k + 1)) < 1e-12
    assert 1e-5 < abs(bh @ c**4 - 1 / 5) < 1e-2


@pytest.mark.parametrize("dims", [(2, 6, 2), (16, 48, 16), (32, 128, 128, 32), (5, 7, 3, 5)])
@pytest.mark.parametrize("act", [O.ACT_TANH, O.ACT_SOFTPLUS, O.ACT_SWISH, O.ACT_SIGMOID, O.ACT_ELU])
def test_vjp_jvp_jacobian_match_autograd(dims, act):
    rng = np.random.default_rng(0)
    net = _net(dims, act)
    flat = O.glorot_params(net, rng, np.float64, bias_scale=0.1)
    B = 5
    z = rng.standard_normal((dims[0], B))
    eps = rng.standard_normal((dims[0], B))
    f = _torch_mlp(net, flat)
    y, eJ = O.mlp_vjp(net, flat, z, eps)
    _, Je = O.mlp_jvp(net, flat, z, eps)
    _, Jv = O.jacobian_batched(net, flat, z, use_jvp=False)
    _, Jf = O.jacobian_batched(net, flat, z, use_jvp=True)
    for b in range(B):
        zb, eb = torch.tensor(z[:, b]), torch.tensor(eps[:, b])
        yt, vjp_fn = torch.func.vjp(f, zb)
        assert np.allclose(y[:, b], yt.numpy(), atol=1e-13)
        assert np.allclose(eJ[:, b], vjp_fn(eb)[0].numpy(), atol=1e-12)
        assert np.allclose(Je[:, b], torch.func.jvp(f, (zb,), (eb,))[1].numpy(), atol=1e-12)
        Jt = torch.func.jacrev(f)(zb).numpy()
        assert np.allclose(Jv[:, :, b], Jt, atol=1e-12)
        assert np.allclose(Jf[:, :, b], Jt, atol=1e-12)


def test_jacobian_finite_differences():
    rng = np.random.default_rng(1)
    net = _net((4, 9, 4))
    flat = O.glorot_params(net, rng, np.float64, bias_scale=0.2)
    z = rng.standard_normal((4, 3))
    _, J = O.jacobian_batched(net, flat, z)
    h = 1e-6
    for i in range(4):
        dz = np.zeros_like(z)
        dz[i] = h
        col = (O.mlp_forward(net, flat, z + dz)[0] - O.mlp_forward(net, flat, z - dz)[0]) / (2 * h)
        assert np.allclose(J[:, i, :], col, atol=1e-8)


def test_augmented_f_rows_and_modes():
    rng = np.random.default_rng(2)
    net = _net((6, 10, 6))
    flat = O.glorot_params(net, rng, np.float64, bias_scale=0.1)
    B, n = 7, 6
    u = rng.standard_normal((n + 3, B))
    eps = rng.standard_normal((n, B))
    dv = O.augmented_f_train(net, flat, u, eps, True, True, use_jvp=False)
    dj = O.augmented_f_train(net, flat, u, eps, True, True, use_jvp=True)
    _, J = O.jacobian_batched(net, flat, u[:n])
    for b in range(B):
        Jb = J[:, :, b]
        assert np.allclose(dv[n, b], -eps[:, b] @ Jb @ eps[:, b], atol=1e-12)      # icnf.jl:334
        assert np.allclose(dj[n, b], dv[n, b], atol=1e-12)                        # same scalar
        assert np.allclose(dv[n + 2, b], np.linalg.norm(Jb.T @ eps[:, b]), atol=1e-12)  # VJP n-row
        assert np.allclose(dj[n + 2, b], np.linalg.norm(Jb @ eps[:, b]), atol=1e-12)    # JVP n-row
        assert np.allclose(dv[n + 1, b], np.linalg.norm(dv[:n, b]), atol=1e-12)
    assert not np.allclose(dv[n + 2], dj[n + 2])     # the two modes are not interchangeable
    # switches off -> zero rows (icnf.jl:337-340, 344-347)
    d0 = O.augmented_f_train(net, flat, u, eps, False, False)
    assert np.all(d0[n + 1:] == 0) and np.allclose(d0[:n + 1], dv[:n + 1])
    # TestMode: exact trace, no regulariser rows; the RHS ignores the non-z rows of u
    dt = O.augmented_f_test(net, flat, u[:n + 1])
    assert dt.shape == (n + 1, B)
    assert np.allclose(dt[n], -np.trace(J, axis1=0, axis2=1), atol=1e-12)
    u2 = u.copy(); u2[n:] += 3.0
    assert np.allclose(O.augmented_f_train(net, flat, u2, eps, True, True), dv)


def test_hutchinson_unbiased():
    rng = np.random.default_rng(3)
    net = _net((4, 8, 4))
    flat = O.glorot_params(net, rng, np.float64, bias_scale=0.1)
    z = rng.standard_normal((4, 1))
    _, J = O.jacobian_batched(net, flat, z)
    tr = np.trace(J[:, :, 0])
    N = 20000
    Z = np.repeat(z, N, axis=1)
    eps = rng.standard_normal((4, N))
    u = np.vstack([Z, np.zeros((3, N))])
    est = -O.augmented_f_train(net, flat, u, eps, False, False)[4]
    assert abs(est.mean() - tr) < 4 * est.std() / math.sqrt(N)


def test_linear_field_closed_form():
    """f(z) = A z: tr J = tr A, z(t1) = expm(A dt) z0, dlogp = -tr(A) dt."""
    rng = np.random.default_rng(4)
    n, B = 5, 6
    A = 0.3 * rng.standard_normal((n, n))
    net = O.Net((n, n), (O.ACT_IDENTITY,))
    flat = np.concatenate([A.T.reshape(-1), np.zeros(n)])   # column-major weight, zero bias
    cfg = O.Cfg(net, n, 0, 0.0, 0.0, 0.0, tspan=(0.0, 1.5))
    xs = rng.standard_normal((n, B))
    eps = rng.standard_normal((n, B))
    # Test mode (exact trace)
    fsol, logpx, regs, st = O.inference(cfg, flat, xs, eps, train=False, dt=1 / 64, adaptive=False)
    zt = scipy.linalg.expm(1.5 * A) @ xs
    assert np.allclose(fsol[:n], zt, atol=1e-9)
    assert np.allclose(fsol[n], -np.trace(A) * 1.5, atol=1e-9)
    ref_lp = scipy.stats.multivariate_normal(np.zeros(n), np.eye(n)).logpdf(zt.T) + np.trace(A) * 1.5
    assert np.allclose(logpx, ref_lp, atol=1e-8)
    # Train mode: Hutchinson value is eps^T A eps, constant in time
    cfg2 = O.Cfg(net, n, 0, 1e-2, 1e-2, 0.0, tspan=(0.0, 1.5))
    fsol2, _, (E, nn_, Aa), _ = O.inference(cfg2, flat, xs, eps, train=True, dt=1 / 64, adaptive=False)
    assert np.allclose(fsol2[n], -1.5 * np.einsum("ib,ij,jb->b", eps, A, eps), atol=1e-9)
    assert np.allclose(nn_, 1.5 * np.linalg.norm(A.T @ eps, axis=0), atol=1e-9)      # VJP: ||A^T eps||
    cfg3 = O.Cfg(net, n, 0, 1e-2, 1e-2, 0.0, use_jvp=True, tspan=(0.0, 1.5))
    _, _, (_, nj, _), _ = O.inference(cfg3, flat, xs, eps, train=True, dt=1 / 64, adaptive=False)
    assert np.allclose(nj, 1.5 * np.linalg.norm(A @ eps, axis=0), atol=1e-9)         # JVP: ||A eps||


def test_tsit5_vs_scipy_and_adaptive():
    cfg, _, train = O.baseline_cfg(2)
    rng = np.random.default_rng(5)
    flat = O.glorot_params(cfg.net, rng, np.float64)
    B = 8
    xs = rng.standard_normal((cfg.nvars, B))
    eps = rng.standard_normal((cfg.n_in, B))
    u0 = O.inference_u0(cfg, xs, train)
    f = cfg.rhs(flat, eps, train)
    ref = scipy.integrate.solve_ivp(lambda t, y: f(y.reshape(u0.shape)).reshape(-1), (0, 1),
                                    u0.reshape(-1), method="DOP853", rtol=1e-12, atol=1e-12)
    yref = ref.y[:, -1].reshape(u0.shape)
    yfix, st = O.tsit5_solve(f, u0, 0.0, 1.0, dt=1 / 64, adaptive=False)
    assert st.naccept == 64 and st.nf == 1 + 6 * 64
    assert np.allclose(yfix, yref, rtol=1e-7, atol=1e-9)
    yad, st2 = O.tsit5_solve(f, u0, 0.0, 1.0, abstol=1e-9, reltol=1e-9)
    assert np.allclose(yad, yref, rtol=1e-6, atol=1e-8)
    assert st2.nf == 2 + 6 * (st2.naccept + st2.nreject)
    # reversed time returns to the start (generate direction)
    yback, _ = O.tsit5_solve(f, yfix, 1.0, 0.0, dt=1 / 64, adaptive=False)
    assert np.allclose(yback[:cfg.n_in], u0[:cfg.n_in], atol=1e-7)


def test_post_and_loss():
    cfg, _, _ = O.baseline_cfg(2)
    rng = np.random.default_rng(6)
    B = 9
    fsol = rng.standard_normal((cfg.D(True), B))
    logpx, (E, n, A) = O.inference_sol(cfg, fsol, True)
    z = fsol[:cfg.n_in]
    ref = scipy.stats.multivariate_normal(np.zeros(cfg.n_in), np.eye(cfg.n_in)).logpdf(z.T) - fsol[cfg.n_in]
    assert np.allclose(logpx, ref, atol=1e-12)
    assert np.allclose(A, np.linalg.norm(z[cfg.nvars:], axis=0))
    assert np.allclose(E, fsol[cfg.n_in + 1]) and np.allclose(n, fsol[cfg.n_in + 2])
    L = O.loss(cfg, logpx, (E, n, A), True)
    assert np.isclose(L, np.mean(-logpx + cfg.lam1 * E + cfg.lam2 * n + cfg.lam3 * A))
    lp2, (e2, n2, a2) = O.inference_sol(cfg, fsol[:cfg.D(False)], False)
    assert e2 is None and n2 is None
    assert np.isclose(O.loss(cfg, lp2, (e2, n2, a2), False), -np.mean(lp2))


def test_param_layout_is_column_major_weight_then_bias():
    net = O.Net((2, 3), (O.ACT_IDENTITY,))
    flat = np.arange(9, dtype=np.float64)
    (W,), (b,) = O.unflatten_params(net, flat)
    assert W.shape == (3, 2)
    assert np.array_equal(W[:, 0], [0, 1, 2]) and np.array_equal(W[:, 1], [3, 4, 5])
    assert np.array_equal(b, [6, 7, 8])


def test_conditional_layer_matches_autograd():
    """CondLayer: nn(vcat(z, ys)) (src/layers/cond_layer.jl:7-9); AD products w.r.t. z only."""
    rng = np.random.default_rng(8)
    n_in, n_cond, B = 4, 3, 5
    net = O.Net((n_in + n_cond, 9, n_in), (O.ACT_TANH, O.ACT_TANH))
    flat = O.glorot_params(net, rng, np.float64, bias_scale=0.1)
    z = rng.standard_normal((n_in, B)); ys = rng.standard_normal((n_cond, B)); eps = rng.standard_normal((n_in, B))
    f = _torch_mlp(net, flat)
    y, eJ = O.mlp_vjp(net, flat, z, eps, ys)
    _, Je = O.mlp_jvp(net, flat, z, eps, ys)
    _, J = O.jacobian_batched(net, flat, z, ys=ys)
    for b in range(B):
        yb = torch.tensor(ys[:, b])
        fz = lambda zz: f(torch.cat([zz, yb]))
        zb, eb = torch.tensor(z[:, b]), torch.tensor(eps[:, b])
        yt, vjp_fn = torch.func.vjp(fz, zb)
        assert np.allclose(y[:, b], yt.numpy(), atol=1e-13)
        assert np.allclose(eJ[:, b], vjp_fn(eb)[0].numpy(), atol=1e-12)
        assert np.allclose(Je[:, b], torch.func.jvp(fz, (zb,), (eb,))[1].numpy(), atol=1e-12)
        assert np.allclose(J[:, :, b], torch.func.jacrev(fz)(zb).numpy(), atol=1e-12)
    # folding the conditioning input into a per-sample bias is the same function
    Ws, bs = O.unflatten_params(net, flat)
    a1 = Ws[0][:, :n_in] @ z + (Ws[0][:, n_in:] @ ys + bs[0][:, None])
    assert np.allclose(O.mlp_forward(net, flat, z, ys)[1][1], np.tanh(a1), atol=1e-14)


def test_blas_baseline_matches_the_oracle():
    """oracle/cnf_blas.py (bench.py's cpu_baseline: sgemm over the whole n x B matrices on torch-CPU) computes the
    same RHS and the same solve as the numpy oracle."""
    from oracle import cnf_blas as BL
    from tests.helpers import assert_parity
    for i in (2, 3):
        cfg, _, _ = O.baseline_cfg(i)
        rng = np.random.default_rng(40 + i)
        flat = O.glorot_params(cfg.net, rng, np.float32, 0.1)
        B = 96
        xs = rng.standard_normal((cfg.nvars, B)).astype(np.float32)
        eps = rng.standard_normal((cfg.n_in, B)).astype(np.float32)
        u0 = O.inference_u0(cfg, xs, True)
        ref = cfg.rhs(flat.astype(np.float64), eps.astype(np.float64), True)(u0.astype(np.float64))
        assert_parity(BL.make_rhs(cfg, flat, eps)(u0), ref, f"blas rhs cfg{i}", trace_row=cfg.n_in)
        got, st = BL.solve(cfg, flat, u0, eps, dt=1 / 8, adaptive=False)
        want, st2 = O.tsit5_solve(cfg.rhs(flat.astype(np.float64), eps.astype(np.float64), True), u0.astype(np.float64),
                                  0.0, 1.0, dt=1 / 8, adaptive=False)
        assert st["nf"] == st2.nf
        assert_parity(got, want, f"blas solve cfg{i}", trace_row=cfg.n_in)
        # the all-torch driver bench.py times (solve_torch): same fixed-dt result, and the same adaptive step sequence
        # as the numpy driver around the same right-hand side
        got_t, st_t = BL.solve_torch(cfg, flat, u0, eps, dt=1 / 8, adaptive=False)
        assert st_t["nf"] == st2.nf
        assert_parity(got_t, want, f"torch-driver solve cfg{i}", trace_row=cfg.n_in)
        kw = dict(reltol=3.45e-4, abstol=1.19e-7)
        ga, sa = BL.solve(cfg, flat, u0, eps, **kw)
        gt, stt = BL.solve_torch(cfg, flat, u0, eps, **kw)
        assert (sa["nf"], sa["naccept"], sa["nreject"]) == (stt["nf"], stt["naccept"], stt["nreject"])
        assert_parity(gt, ga, f"torch-driver adaptive cfg{i}", rtol=1e-3, trace_row=cfg.n_in)
