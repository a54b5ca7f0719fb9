"""Pins oracle/cnf_grad_oracle.py (row f3: d loss / d ps through the Tsit5 solve) against torch's
reverse-mode autograd of the same discrete computation in float64, and against finite differences."""
import numpy as np
import pytest
import torch

from oracle import cnf_grad_oracle as G
from oracle import cnf_oracle as O

ACT_T = {O.ACT_IDENTITY: lambda a: a, O.ACT_TANH: torch.tanh, O.ACT_SIGMOID: torch.sigmoid,
         O.ACT_SOFTPLUS: torch.nn.functional.softplus, O.ACT_SWISH: torch.nn.functional.silu,
         O.ACT_ELU: torch.nn.functional.elu}


def _torch_net(net, flat, ys):
    Ws, bs, off = [], [], 0
    for i, o in zip(net.dims[:-1], net.dims[1:]):
        Ws.append(flat[off:off + i * o].reshape(i, o).T)
        off += i * o
        bs.append(flat[off:off + o])
        off += o

    def nn(z):                       # z: n_in x B
        h = z if ys is None else torch.cat([z, ys], 0)
        for W, b, k in zip(Ws, bs, net.acts):
            h = ACT_T[k](W @ h + b[:, None])
        return h
    return nn


def _torch_rhs(cfg, flat, eps, ys):
    nn = _torch_net(cfg.net, flat, ys)
    n_in = cfg.n_in

    def f(u):
        z = u[:n_in]
        if cfg.use_jvp:
            zdot, eJ = torch.func.jvp(nn, (z,), (eps,))
        else:
            zdot, pull = torch.func.vjp(nn, z)
            eJ, = pull(eps)
        ldot = -(eJ * eps).sum(0, keepdim=True)
        Ed = zdot.norm(dim=0, keepdim=True) if cfg.lam1 != 0 else torch.zeros_like(ldot)
        nd = eJ.norm(dim=0, keepdim=True) if cfg.lam2 != 0 else torch.zeros_like(ldot)
        return torch.cat([zdot, ldot, Ed, nd], 0)
    return f


def _torch_loss(cfg, flat, xs, eps, ys, dts):
    f = _torch_rhs(cfg, flat, eps, ys)
    B = xs.shape[1]
    u = torch.cat([xs, torch.zeros(cfg.naugs + 3, B, dtype=xs.dtype)], 0)
    tdir = 1.0 if cfg.tspan[1] >= cfg.tspan[0] else -1.0
    for h in dts:
        h = tdir * h
        ks = []
        for s in range(6):
            acc = torch.zeros_like(u)
            for j in range(s):
                acc = acc + O.TSIT5_A[s][j] * ks[j]
            ks.append(f(u + h * acc))
        acc = torch.zeros_like(u)
        for j in range(6):
            acc = acc + O.TSIT5_B[j] * ks[j]
        u = u + h * acc
    n_in = cfg.n_in
    z = u[:n_in]
    logpz = -0.5 * (n_in * np.log(2 * np.pi) + (z * z).sum(0))
    logpx = logpz - u[n_in]
    A = z[cfg.nvars:].norm(dim=0) if (cfg.lam3 != 0 and cfg.naugs > 0) else torch.zeros(B, dtype=xs.dtype)
    return (-logpx + cfg.lam1 * u[n_in + 1] + cfg.lam2 * u[n_in + 2] + cfg.lam3 * A).mean()


CASES = [
    dict(dims=(4, 9, 4), acts=(O.ACT_TANH, O.ACT_TANH), nvars=2, naugs=2, lam=(1e-2, 1e-2, 1e-2), jvp=False),
    dict(dims=(3, 8, 5, 3), acts=(O.ACT_TANH, O.ACT_SOFTPLUS, O.ACT_IDENTITY), nvars=3, naugs=0,
         lam=(1e-2, 1e-2, 0.0), jvp=False),
    dict(dims=(4, 7, 4), acts=(O.ACT_SWISH, O.ACT_SIGMOID), nvars=3, naugs=1, lam=(0.0, 0.0, 0.0), jvp=False),
    dict(dims=(4, 9, 4), acts=(O.ACT_TANH, O.ACT_TANH), nvars=2, naugs=2, lam=(1e-2, 1e-2, 1e-2), jvp=True),
    dict(dims=(5, 6, 3), acts=(O.ACT_ELU, O.ACT_TANH), nvars=3, naugs=0, lam=(1e-2, 1e-2, 0.0), jvp=False, ncond=2),
]


@pytest.mark.parametrize("case", range(len(CASES)))
@pytest.mark.parametrize("adaptive", [False, True])
def test_grad_matches_torch_autograd(case, adaptive):
    c = CASES[case]
    cfg = O.Cfg(O.Net(c["dims"], c["acts"]), c["nvars"], c["naugs"], *c["lam"], use_jvp=c["jvp"], tspan=(0.0, 1.0))
    rng = np.random.default_rng(100 + case)
    B = 7
    flat = O.glorot_params(cfg.net, rng, np.float64, 0.2)
    xs = rng.standard_normal((cfg.nvars, B))
    eps = rng.standard_normal((cfg.n_in, B))
    ys = rng.standard_normal((c["ncond"], B)) if c.get("ncond") else None
    kw = dict(adaptive=True, reltol=1e-5, abstol=1e-7) if adaptive else dict(adaptive=False, dt=0.25)
    val, grad, st = G.loss_and_grad(cfg, flat, xs, eps, ys, **kw)
    ft = torch.tensor(flat, requires_grad=True)
    lt = _torch_loss(cfg, ft, torch.tensor(xs), torch.tensor(eps), None if ys is None else torch.tensor(ys), st.dts)
    lt.backward()
    assert abs(val - float(lt)) <= 1e-12 * max(1.0, abs(val))
    ref = ft.grad.numpy()
    assert np.abs(grad - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max()), np.abs(grad - ref).max()


def test_grad_matches_finite_differences():
    c = CASES[0]
    cfg = O.Cfg(O.Net(c["dims"], c["acts"]), c["nvars"], c["naugs"], *c["lam"], tspan=(0.0, 1.0))
    rng = np.random.default_rng(3)
    B = 5
    flat = O.glorot_params(cfg.net, rng, np.float64, 0.2)
    xs = rng.standard_normal((cfg.nvars, B))
    eps = rng.standard_normal((cfg.n_in, B))
    kw = dict(adaptive=False, dt=0.125)
    _, grad, _ = G.loss_and_grad(cfg, flat, xs, eps, **kw)

    def L(p):
        _, logpx, regs, _ = O.inference(cfg, p, xs, eps, True, **kw)
        return O.loss(cfg, logpx, regs, True)
    for i in rng.choice(flat.size, 12, replace=False):
        e = np.zeros_like(flat)
        e[i] = 1e-6
        fd = (L(flat + e) - L(flat - e)) / 2e-6
        assert abs(fd - grad[i]) <= 1e-6 * max(1.0, abs(grad[i])), (i, fd, grad[i])
    # the gradient w.r.t. the data (st.grad_x: the adjoint state at t0, rows of xs) against finite differences, for every
    # case shape (augmented rows included: they are zeros in u0 and have no entry in grad_x)
    for c in CASES:
        cfg = O.Cfg(O.Net(c["dims"], c["acts"]), c["nvars"], c["naugs"], *c["lam"], use_jvp=c["jvp"], tspan=(0.0, 1.0))
        flat = O.glorot_params(cfg.net, rng, np.float64, 0.2)
        xs = rng.standard_normal((cfg.nvars, B))
        eps = rng.standard_normal((cfg.n_in, B))
        ys = rng.standard_normal((c["ncond"], B)) if c.get("ncond") else None
        _, _, st = G.loss_and_grad(cfg, flat, xs, eps, ys, **kw)
        assert st.grad_x.shape == xs.shape

        def Lx(x):
            _, logpx, regs, _ = O.inference(cfg, flat, x, eps, True, ys, **kw)
            return O.loss(cfg, logpx, regs, True)
        for _ in range(6):
            i, b = rng.integers(cfg.nvars), rng.integers(B)
            e = np.zeros_like(xs)
            e[i, b] = 1e-6
            fd = (Lx(xs + e) - Lx(xs - e)) / 2e-6
            assert abs(fd - st.grad_x[i, b]) <= 1e-6 * max(1.0, abs(fd)), (c["dims"], i, b, fd, st.grad_x[i, b])


def test_rhs_vjp_matches_autograd_on_baseline_shapes():
    """The pullback of one augmented_f evaluation on the BASELINE network shapes."""
    for i in (1, 2, 3):
        cfg, _, _ = O.baseline_cfg(i)
        rng = np.random.default_rng(40 + i)
        B = 6
        flat = O.glorot_params(cfg.net, rng, np.float64, 0.1)
        u = rng.standard_normal((cfg.D(True), B))
        eps = rng.standard_normal((cfg.n_in, B))
        cot = rng.standard_normal((cfg.D(True), B))
        zbar, g = G.rhs_vjp(cfg.net, flat, u[:cfg.n_in], eps, cot, cfg.lam1 != 0, cfg.lam2 != 0)
        ft = torch.tensor(flat, requires_grad=True)
        ut = torch.tensor(u, requires_grad=True)
        out = _torch_rhs(cfg, ft, torch.tensor(eps), None)(ut)
        (out * torch.tensor(cot)).sum().backward()
        assert np.abs(g - ft.grad.numpy()).max() <= 1e-10 * max(1.0, np.abs(g).max())
        assert np.abs(zbar - ut.grad.numpy()[:cfg.n_in]).max() <= 1e-10
        assert np.abs(ut.grad.numpy()[cfg.n_in:]).max() == 0


# ---- TestMode (exact trace): oracle/cnf_grad_oracle.py::loss_and_grad_test against torch autograd -------------------
def _torch_loss_test(cfg, flat, xs, ys, dts):
    """-mean(logpx) of the TestMode solve replayed on the steps ``dts``: the Jacobian of every column is formed layer by layer
    (J = D_L W_L ... D_1 W_1) and its trace taken -- plain differentiable tensor operations."""
    net, n_in = cfg.net, cfg.n_in
    Ws, bs, off = [], [], 0
    for i, o in zip(net.dims[:-1], net.dims[1:]):
        Ws.append(flat[off:off + i * o].reshape(i, o).T); off += i * o
        bs.append(flat[off:off + o]); off += o
    D1 = {O.ACT_IDENTITY: lambda a: torch.ones_like(a), O.ACT_TANH: lambda a: 1 - torch.tanh(a) ** 2,
          O.ACT_SIGMOID: lambda a: torch.sigmoid(a) * (1 - torch.sigmoid(a)), O.ACT_SOFTPLUS: torch.sigmoid}

    def f(u):
        z = u[:n_in]
        h = z if ys is None else torch.cat([z, ys], 0)
        Bn = z.shape[1]
        J = torch.eye(h.shape[0], n_in, dtype=u.dtype).expand(Bn, h.shape[0], n_in)
        for W, b, k in zip(Ws, bs, net.acts):
            a = W @ h + b[:, None]
            J = D1[k](a).T[:, :, None] * (W[None] @ J)
            h = ACT_T[k](a)
        tr = torch.diagonal(J, dim1=1, dim2=2).sum(1)
        return torch.cat([h, -tr[None, :]], 0)
    B = xs.shape[1]
    u = torch.cat([xs, torch.zeros(cfg.naugs + 1, B, dtype=xs.dtype)], 0)
    tdir = 1.0 if cfg.tspan[1] >= cfg.tspan[0] else -1.0
    for h in dts:
        h = tdir * h
        ks = []
        for s in range(6):
            acc = torch.zeros_like(u)
            for j in range(s):
                acc = acc + O.TSIT5_A[s][j] * ks[j]
            ks.append(f(u + h * acc))
        acc = torch.zeros_like(u)
        for j in range(6):
            acc = acc + O.TSIT5_B[j] * ks[j]
        u = u + h * acc
    z = u[:n_in]
    logpx = -0.5 * (n_in * np.log(2 * np.pi) + (z * z).sum(0)) - u[n_in]
    return (-logpx).mean()


TEST_CASES = [
    dict(dims=(4, 9, 4), acts=(O.ACT_TANH, O.ACT_TANH), nvars=2, naugs=2),
    dict(dims=(5, 5), acts=(O.ACT_TANH,), nvars=3, naugs=2),                                 # the benchmark suite's one-layer form
    dict(dims=(3, 8, 5, 3), acts=(O.ACT_TANH, O.ACT_SOFTPLUS, O.ACT_IDENTITY), nvars=3, naugs=0),
    dict(dims=(5, 6, 3), acts=(O.ACT_SIGMOID, O.ACT_TANH), nvars=3, naugs=0, ncond=2),
]


@pytest.mark.parametrize("case", range(len(TEST_CASES)))
def test_testmode_grad_matches_torch_autograd(case):
    c = TEST_CASES[case]
    cfg = O.Cfg(O.Net(c["dims"], c["acts"]), c["nvars"], c["naugs"], 1e-2, 1e-2, 1e-2, tspan=(0.0, 1.0))
    rng = np.random.default_rng(200 + case)
    B = 6
    flat = O.glorot_params(cfg.net, rng, np.float64, 0.4)
    flat[-cfg.n_in:] = 0.1 * rng.standard_normal(cfg.n_in)
    xs = rng.standard_normal((cfg.nvars, B))
    ys = rng.standard_normal((c["ncond"], B)) if c.get("ncond") else None
    val, grad, st = G.loss_and_grad_test(cfg, flat, xs, ys, adaptive=True, reltol=1e-5, abstol=1e-7)
    ft = torch.tensor(flat, requires_grad=True)
    xt = torch.tensor(xs, requires_grad=True)
    lt = _torch_loss_test(cfg, ft, xt, None if ys is None else torch.tensor(ys), st.dts)
    lt.backward()
    assert abs(val - float(lt)) <= 1e-12 * max(1.0, abs(val))
    ref = ft.grad.numpy()
    assert np.abs(grad - ref).max() <= 1e-10 * max(1.0, np.abs(ref).max()), np.abs(grad - ref).max()
    assert np.abs(st.grad_x - xt.grad.numpy()).max() <= 1e-10 * max(1.0, np.abs(xt.grad.numpy()).max())
