"""CPU oracle for the gradient of the training loss w.r.t. the flat parameter vector
(SURVEY.md section 8(f) row f3).

TEST INFRASTRUCTURE ONLY -- see the header of cnf_oracle.py; the same rules apply.
PARITY UNPINNED against the Julia package (it cannot run here); pinned instead by reverse-mode
autograd of the SAME discrete computation (torch, float64) and by finite differences
(tests/test_grad_oracle.py).

What the reference does: ``MLJModelInterface.fit`` (src/exts/mlj_ext/core_icnf.jl:59-73) hands
``loss(icnf, TrainMode(), xs, ps, st)`` (src/icnf.jl:481-490) to Optimization.jl with
``AutoEnzyme``; the derivative goes through ``solve`` by SciMLSensitivity's adjoint
(Project.toml:31; third party, not vendored).  Restated here as the DISCRETE adjoint of the Tsit5
steps actually taken (step sizes are constants of the differentiation, as in every ODE adjoint):
it is the exact gradient of the number ``loss`` returns, so it can be checked to rounding.

Per step  u+ = u + h sum_i b_i k_i,  k_i = f(U_i),  U_i = u + h sum_{j<i} a_ij k_j  (k_1 = f(u):
FSAL only saves the evaluation, the function of u is the same):
    for i = 6..1:  kbar_i = h (b_i lam + sum_{m>i} a_mi w_m);   (w_i, g_i) = vjp of f at U_i with kbar_i
    lam <- lam + sum_i w_i;   grad += sum_i g_i

vjp of the augmented right-hand side (src/icnf.jl:318-350).  Only the z rows of u enter f, so with
the cotangent (a, c_l, c_E, c_n) of (zdot, ldot, Edot, ndot) the scalar to differentiate is
    Phi(z, theta) = ahat' nn(z) + omega' J(z) tau
    VJP mode:  ahat = a + c_E zdot/|zdot|,  omega = eps,  tau = -c_l eps + c_n eJ/|eJ|   (eJ = J' eps)
    JVP mode:  ahat as above,  omega = -c_l eps + c_n Je/|Je|,  tau = eps                  (Je = J eps)
(the unit vectors are evaluated at the point: first-order chain rule).  For the MLP
h_l = s(a_l), a_l = W_l h_{l-1} + b_l, tangent t_l = s'(a_l) .* (W_l t_{l-1}), t_0 = tau:
    Phi = ahat' h_L + omega' t_L
    hbar_L = ahat, tbar_L = omega;  for l = L..1 with p_l = W_l t_{l-1}:
        abar_l = hbar_l .* s'(a_l) + tbar_l .* s''(a_l) .* p_l
        pbar_l = tbar_l .* s'(a_l)
        Wbar_l += abar_l h_{l-1}' + pbar_l t_{l-1}';   bbar_l += abar_l
        hbar_{l-1} = W_l' abar_l;   tbar_{l-1} = W_l' pbar_l
    zbar = hbar_0 (rows of z).
"""
from __future__ import annotations

import numpy as np

from . import cnf_oracle as O


def act_d2(kind: int, a: np.ndarray):
    """Second derivative of the activation w.r.t. its pre-activation."""
    one = a.dtype.type(1)
    if kind in (O.ACT_IDENTITY, O.ACT_RELU):
        return np.zeros_like(a)
    if kind == O.ACT_TANH:
        h = np.tanh(a)
        return -2 * h * (one - h * h)
    if kind == O.ACT_SIGMOID:
        s = O._sigmoid(a)
        return s * (one - s) * (one - 2 * s)
    if kind == O.ACT_SOFTPLUS:
        s = O._sigmoid(a)
        return s * (one - s)
    if kind == O.ACT_SWISH:
        s = O._sigmoid(a)
        ds = s * (one - s)
        return 2 * ds + a * ds * (one - 2 * s)
    if kind == O.ACT_ELU:
        return np.where(a > 0, np.zeros_like(a), np.exp(np.minimum(a, 0)))
    raise ValueError(kind)


def _unit(v):
    n = np.sqrt(np.sum(v * v, axis=0, keepdims=True))
    return np.divide(v, n, out=np.zeros_like(v), where=n > 0)


def flatten_grads(net: O.Net, gWs, gbs):
    parts = []
    for gW, gb in zip(gWs, gbs):
        parts.append(np.asarray(gW).T.reshape(-1))      # out x in, column-major
        parts.append(np.asarray(gb).reshape(-1))
    return np.concatenate(parts)


def rhs_vjp(net: O.Net, flat, z, eps, cot, norm_z: bool, norm_j: bool, use_jvp=False, ys=None):
    """Pullback of augmented_f (TrainMode) at z.  ``cot``: D x B cotangent of du = [zdot; ldot; Edot;
    ndot].  Returns (zbar [n_in x B], grad [n_params]) -- grad summed over the columns."""
    n_in = z.shape[0]
    Ws, bs = O.unflatten_params(net, flat)
    x0 = z if ys is None else np.vstack([z, ys])
    # forward with pre-activations
    hs, as_ = [x0], []
    h = x0
    for W, b in zip(Ws, bs):
        a = W @ h + b[:, None]
        as_.append(a)
        h = O.act_apply(net.acts[len(as_) - 1], a)[0]
        hs.append(h)
    d1 = [O.act_apply(k, a)[1] for k, a in zip(net.acts, as_)]
    d2 = [act_d2(k, a) for k, a in zip(net.acts, as_)]
    zdot = hs[-1]
    a_z, c_l, c_E, c_n = cot[:n_in], cot[n_in:n_in + 1], cot[n_in + 1:n_in + 2], cot[n_in + 2:n_in + 3]
    ahat = a_z + (c_E * _unit(zdot) if norm_z else 0)
    if use_jvp:
        _, Je = O.mlp_jvp(net, flat, z, eps, ys)
        omega = -c_l * eps + (c_n * _unit(Je) if norm_j else 0)
        tau = eps
    else:
        _, eJ = O.mlp_vjp(net, flat, z, eps, ys)
        omega = eps
        tau = -c_l * eps + (c_n * _unit(eJ) if norm_j else 0)
    # forward tangent sweep
    t = tau if ys is None else np.vstack([tau, np.zeros_like(ys)])
    ts, ps = [t], []
    for W, d in zip(Ws, d1):
        p = W @ t
        ps.append(p)
        t = d * p
        ts.append(t)
    # reverse sweep
    hbar, tbar = ahat, omega
    gWs, gbs = [None] * len(Ws), [None] * len(Ws)
    for l in reversed(range(len(Ws))):
        abar = hbar * d1[l] + tbar * d2[l] * ps[l]
        pbar = tbar * d1[l]
        gWs[l] = abar @ hs[l].T + pbar @ ts[l].T
        gbs[l] = abar.sum(axis=1)
        hbar = Ws[l].T @ abar
        tbar = Ws[l].T @ pbar
    return hbar[:n_in], flatten_grads(net, gWs, gbs)


def forward_record(f, u0, t0, t1, dts):
    """Replays the accepted steps ``dts`` (absolute sizes) and returns the states [u_0 .. u_N]."""
    T = u0.dtype.type
    tdir = T(1) if t1 >= t0 else T(-1)
    us = [u0.copy()]
    u, k1 = u0, f(u0)
    for h in dts:
        u, k1, _ = O.tsit5_step(f, u, k1, tdir * T(h))
        us.append(u)
    return us


def final_cotangent(cfg: O.Cfg, fsol):
    """d loss / d fsol for loss = mean_b(-logpx + lam1 E + lam2 n + lam3 A) (src/icnf.jl:481-490 with
    inference_sol src/base_icnf.jl:167-189): -logpx = |z|^2/2 + const + dlogp."""
    n_in, B = cfg.n_in, fsol.shape[1]
    lam = np.zeros_like(fsol)
    z = fsol[:n_in]
    lam[:n_in] = z
    if cfg.lam3 != 0 and cfg.naugs > 0:
        lam[cfg.nvars:n_in] += cfg.lam3 * _unit(z[cfg.nvars:])
    lam[n_in] = 1
    lam[n_in + 1] = cfg.lam1
    lam[n_in + 2] = cfg.lam2
    return lam / B


def loss_and_grad(cfg: O.Cfg, flat, xs, eps, ys=None, dts=None, **solve_kw):
    """(loss, d loss / d flat, stats) of the TrainMode loss through the Tsit5 solve.  ``dts``: replay
    these accepted step sizes instead of solving (to differentiate exactly the discrete map another
    implementation took)."""
    flat = np.asarray(flat)
    u0 = O.inference_u0(cfg, xs, True)
    f = cfg.rhs(flat, eps, True, ys)
    if dts is None:
        fsol, st = O.tsit5_solve(f, u0, cfg.tspan[0], cfg.tspan[1], **solve_kw)
        us = forward_record(f, u0, cfg.tspan[0], cfg.tspan[1], st.dts)
        assert np.array_equal(us[-1], fsol)
    else:
        st = O.SolveStats(naccept=len(dts), dts=[abs(float(d)) for d in dts])
        us = forward_record(f, u0, cfg.tspan[0], cfg.tspan[1], st.dts)
        fsol = us[-1]
    logpx, regs = O.inference_sol(cfg, fsol, True)
    val = O.loss(cfg, logpx, regs, True)
    T = u0.dtype.type
    tdir = 1.0 if cfg.tspan[1] >= cfg.tspan[0] else -1.0
    n_in = cfg.n_in
    lam = final_cotangent(cfg, fsol)
    grad = np.zeros(flat.size, dtype=flat.dtype)
    A, Bc = O.TSIT5_A, O.TSIT5_B
    nz, nj = cfg.lam1 != 0, cfg.lam2 != 0
    for n in reversed(range(len(st.dts))):
        h = T(tdir * st.dts[n])
        u = us[n]
        ks, Us = [], []
        for s in range(6):
            acc = np.zeros_like(u)
            for j in range(s):
                acc = acc + T(A[s][j]) * ks[j]
            U = u + h * acc
            Us.append(U)
            ks.append(f(U))
        ws = [None] * 6
        for i in reversed(range(6)):
            kbar = T(Bc[i]) * lam
            for m in range(i + 1, 6):
                kbar[:n_in] += T(A[m][i]) * ws[m]
            kbar = h * kbar
            zbar, g = rhs_vjp(cfg.net, flat, Us[i][:n_in], eps, kbar, nz, nj, cfg.use_jvp, ys)
            ws[i] = zbar
            grad += g
        lam = lam.copy()
        for i in range(6):
            lam[:n_in] += ws[i]
    # lam is now d loss / d u(t0); u0 = (xs; zeros): its first nvars rows are d loss / d xs (what the reference's call
    # tests differentiate besides ps: test/call_tests.jl `diff2_loss`)
    st.grad_x = lam[:cfg.nvars].copy()
    return val, grad, st


# ---------------------------------------------------------------------------------------
# TestMode (exact trace): the gradient of ``loss(icnf, TestMode(), xs, ps, st)`` = -mean(logpx) (src/base_icnf.jl:489-497),
# which the reference differentiates in its call tests and in its benchmark suite (test/call_tests.jl `diff_loss` for
# omode = TestMode(); benchmark/benchmarks.jl:60-99 "AD-1-order"/"test") by running Enzyme through ``jacobian_batched``
# (src/utils.jl:1-36) and the solve.  Restated as the same discrete adjoint with the pullback of
#     f(z) = (nn(z), -tr J(z)),      J = D_L W_L ... D_1 W_1,  D_l = diag(s'(a_l))
# written out by hand: with P_l = M_{l-1} ... M_1 (P_1 = [I; 0]), Q_l = M_L ... M_{l+1}, M_l = D_l W_l and G_l = (P_l Q_l)',
#     d tr / d W_l = D_l G_l,      d tr / d a_l (direct) = s''(a_l) .* rowsum(W_l .* G_l),
# and the indirect dependence through h_{l-1} by ordinary back-propagation.  Pinned by torch autograd
# (tests/test_grad_oracle.py::test_testmode_grad_matches_torch_autograd).
# ---------------------------------------------------------------------------------------
def rhs_vjp_test(net: O.Net, flat, z, kbar_z, c, ys=None):
    """Pullback of augmented_f (TestMode) at z: cotangent ``kbar_z`` [n_in x B] of zdot and ``c`` [1 x B] (or scalar) of
    ldot = -tr J.  Returns (zbar [n_in x B], grad [n_params] summed over the columns)."""
    n_in, B = z.shape
    Ws, bs = O.unflatten_params(net, flat)
    L = len(Ws)
    x0 = z if ys is None else np.vstack([z, ys])
    hs, d1, d2 = [x0], [], []
    h = x0
    for l, (W, b) in enumerate(zip(Ws, bs)):
        a = W @ h + b[:, None]
        h, d = O.act_apply(net.acts[l], a)
        hs.append(h); d1.append(d); d2.append(act_d2(net.acts[l], a))
    # per-sample M_l = D_l W_l  [B, out, in]
    Ms = [d1[l].T[:, :, None] * Ws[l][None, :, :] for l in range(L)]
    P = [None] * L
    P[0] = np.broadcast_to(np.eye(x0.shape[0], n_in, dtype=z.dtype), (B, x0.shape[0], n_in))
    for l in range(1, L):
        P[l] = Ms[l - 1] @ P[l - 1]
    Q = [None] * L
    Q[L - 1] = np.broadcast_to(np.eye(n_in, dtype=z.dtype), (B, n_in, n_in))
    for l in range(L - 2, -1, -1):
        Q[l] = Q[l + 1] @ Ms[l + 1]
    cc = np.broadcast_to(np.asarray(c, dtype=z.dtype).reshape(1, -1), (1, B))[0]      # cotangent of ldot per sample
    hbar = kbar_z
    gWs, gbs = [None] * L, [None] * L
    for l in reversed(range(L)):
        G = np.transpose(P[l] @ Q[l], (0, 2, 1))                       # [B, out_l, in_l]
        u = np.einsum("jk,bjk->jb", Ws[l], G)                          # rowsum(W_l .* G_l)
        # Phi = kbar_z' h_L + c * ldot = kbar_z' h_L - c tr J
        abar = hbar * d1[l] + cc[None, :] * (-1.0) * d2[l] * u
        gWs[l] = abar @ hs[l].T - np.einsum("b,jb,bjk->jk", cc, d1[l], G)
        gbs[l] = abar.sum(axis=1)
        hbar = Ws[l].T @ abar
    return hbar[:n_in], flatten_grads(net, gWs, gbs)


def loss_and_grad_test(cfg: O.Cfg, flat, xs, ys=None, dts=None, **solve_kw):
    """(loss, d loss / d flat, stats) of the TestMode loss -mean(logpx) through the Tsit5 solve of the (n_in + 1)-row state
    (exact trace); ``dts`` as in loss_and_grad; ``stats.grad_x`` = d loss / d xs."""
    flat = np.asarray(flat)
    u0 = O.inference_u0(cfg, xs, False)
    f = cfg.rhs(flat, None, False, ys)
    if dts is None:
        fsol, st = O.tsit5_solve(f, u0, cfg.tspan[0], cfg.tspan[1], **solve_kw)
    else:
        st = O.SolveStats(naccept=len(dts), dts=[abs(float(d)) for d in dts])
    us = forward_record(f, u0, cfg.tspan[0], cfg.tspan[1], st.dts)
    fsol = us[-1]
    logpx, regs = O.inference_sol(cfg, fsol, False)
    val = O.loss(cfg, logpx, regs, False)
    T = u0.dtype.type
    tdir = 1.0 if cfg.tspan[1] >= cfg.tspan[0] else -1.0
    n_in, B = cfg.n_in, xs.shape[1]
    lam = fsol[:n_in] / B                                                # -logpx = |z|^2 / 2 + const + dlogp
    lam_l = 1.0 / B
    grad = np.zeros(flat.size, dtype=flat.dtype)
    A, Bc = O.TSIT5_A, O.TSIT5_B
    for n in reversed(range(len(st.dts))):
        h = T(tdir * st.dts[n])
        u = us[n]
        ks, Us = [], []
        for s in range(6):
            acc = np.zeros_like(u)
            for j in range(s):
                acc = acc + T(A[s][j]) * ks[j]
            U = u + h * acc
            Us.append(U)
            ks.append(f(U))
        ws = [None] * 6
        for i in reversed(range(6)):
            kb = T(Bc[i]) * lam
            for m in range(i + 1, 6):
                kb = kb + T(A[m][i]) * ws[m]
            zbar, g = rhs_vjp_test(cfg.net, flat, Us[i][:n_in], h * kb, h * T(Bc[i]) * lam_l, ys)
            ws[i] = zbar
            grad += g
        lam = lam + sum(ws)
    st.grad_x = lam[:cfg.nvars].copy()
    return val, grad, st
