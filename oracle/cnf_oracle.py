"""CPU oracle for the batched augmented-ODE right-hand side of ContinuousNormalizingFlows.jl.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product path.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it, and only as the checker.  The product path (``continuousnf.jl_amd``) never imports it
and fails loudly when the HIP library is missing.

PARITY UNPINNED.  The reference is pure Julia, Julia is not installed in the build
container, and the reference's own tests hold no numeric vectors for this path (every
assertion is ``!isnothing``).  This restatement is therefore pinned
by independent known-answer checks only (tests/test_oracle.py): torch.func vjp/jvp/jacrev,
finite differences, the closed-form linear field, scipy's MvNormal and solve_ivp.

All citations are relative to /root/reference.  Arrays use the reference's logical
shapes: ``u`` is ``D x B`` (rows = state components, columns = samples).  numpy arrays
here are C-ordered ``(D, B)``; the byte layout handed to the C ABI is the Julia
column-major one, i.e. ``u.T`` made contiguous.

The Dense layer, the AD sweeps and Tsit5 are third-party in the reference (Lux, Enzyme,
OrdinaryDiffEq; not vendored) and are restated from their mathematical definitions
(SURVEY.md Appendix A/B).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Sequence

import numpy as np

# --------------------------------------------------------------------------------------
# activations (Lux `Dense(in => out, act)`; reference configs use tanh on every layer,
# README.md:47, test/regression_tests.jl:7)
# --------------------------------------------------------------------------------------
ACT_IDENTITY, ACT_TANH, ACT_SIGMOID, ACT_SOFTPLUS, ACT_RELU, ACT_SWISH, ACT_ELU = range(7)
ACT_NAMES = {
    "identity": ACT_IDENTITY, "tanh": ACT_TANH, "sigmoid": ACT_SIGMOID,
    "softplus": ACT_SOFTPLUS, "relu": ACT_RELU, "swish": ACT_SWISH, "elu": ACT_ELU,
}


def _sigmoid(a):
    return 1.0 / (1.0 + np.exp(-a))


def act_apply(kind: int, a: np.ndarray):
    """Return (h, dh/da) for pre-activation ``a``."""
    one = a.dtype.type(1)
    if kind == ACT_IDENTITY:
        return a, np.ones_like(a)
    if kind == ACT_TANH:
        h = np.tanh(a)
        return h, one - h * h
    if kind == ACT_SIGMOID:
        s = _sigmoid(a)
        return s, s * (one - s)
    if kind == ACT_SOFTPLUS:
        h = np.logaddexp(a, a.dtype.type(0))
        return h, _sigmoid(a)
    if kind == ACT_RELU:
        return np.maximum(a, 0), (a > 0).astype(a.dtype)
    if kind == ACT_SWISH:
        s = _sigmoid(a)
        return a * s, s * (one + a * (one - s))
    if kind == ACT_ELU:
        e = np.exp(np.minimum(a, 0))
        return np.where(a > 0, a, e - one), np.where(a > 0, one, e)
    raise ValueError(f"unknown activation {kind}")


# --------------------------------------------------------------------------------------
# network description + flat parameters
# --------------------------------------------------------------------------------------
@dataclass(frozen=True)
class Net:
    """``Lux.Chain`` of ``Dense`` layers: dims = (n_in, h1, ..., n_out), acts per layer."""
    dims: tuple
    acts: tuple

    @property
    def n_layers(self):
        return len(self.dims) - 1

    @property
    def n_params(self):
        return sum(i * o + o for i, o in zip(self.dims[:-1], self.dims[1:]))


def unflatten_params(net: Net, flat: np.ndarray):
    """Flat layout = ``ComponentArray(Lux.setup(rng, nn)[1])``: per layer ``weight``
    (out x in, column-major) then ``bias`` (out), layers in order (SURVEY.md 8b')."""
    flat = np.asarray(flat)
    assert flat.size == net.n_params, (flat.size, net.n_params)
    Ws, bs, off = [], [], 0
    for i, o in zip(net.dims[:-1], net.dims[1:]):
        Ws.append(flat[off:off + i * o].reshape(i, o).T)  # column-major out x in
        off += i * o
        bs.append(flat[off:off + o])
        off += o
    return Ws, bs


def glorot_params(net: Net, rng: np.random.Generator, dtype=np.float32, bias_scale=0.0):
    """Glorot-uniform weights; biases zero unless bias_scale>0."""
    parts = []
    for i, o in zip(net.dims[:-1], net.dims[1:]):
        lim = math.sqrt(6.0 / (i + o))
        parts.append(rng.uniform(-lim, lim, size=i * o))
        parts.append(bias_scale * rng.standard_normal(o))
    return np.concatenate(parts).astype(dtype)


# --------------------------------------------------------------------------------------
# vector field + AD sweeps (a5)
# --------------------------------------------------------------------------------------
def mlp_forward(net: Net, flat, z, ys=None):
    """``snn(z)`` (src/icnf.jl:329,331): returns (zdot, [h_0..h_L], [sigma'_1..sigma'_L]).
    With ``ys`` the network is the reference's ``CondLayer``: ``nn(vcat(z, ys))``
    (src/layers/cond_layer.jl:7-9)."""
    Ws, bs = unflatten_params(net, flat)
    if ys is not None:
        z = np.vstack([z, ys])
    hs, ds = [z], []
    h = z
    for W, b, k in zip(Ws, bs, net.acts):
        a = W @ h + b[:, None]
        h, d = act_apply(k, a)
        hs.append(h)
        ds.append(d)
    return h, hs, ds


def mlp_vjp(net: Net, flat, z, ct, ys=None):
    """(nn(z), J^T ct): what ``value_and_pullback`` returns at src/icnf.jl:331-332.  With
    ``ys`` the pullback is w.r.t. z only (CondLayer closes over ys)."""
    Ws, _ = unflatten_params(net, flat)
    y, _, ds = mlp_forward(net, flat, z, ys)
    g = ct
    for W, d in zip(reversed(Ws), reversed(ds)):
        g = W.T @ (g * d)
    return y, g[: z.shape[0]]


def mlp_jvp(net: Net, flat, z, tg, ys=None):
    """(nn(z), J tg): what ``value_and_pushforward`` returns at src/icnf.jl:397-402."""
    Ws, _ = unflatten_params(net, flat)
    y, _, ds = mlp_forward(net, flat, z, ys)
    t = tg if ys is None else np.vstack([tg, np.zeros_like(ys)])
    for W, d in zip(Ws, ds):
        t = d * (W @ t)
    return y, t


def jacobian_batched(net: Net, flat, xs, use_jvp=False, ys=None):
    """src/utils.jl:1-17 (VJP rows) / :19-36 (JVP columns): returns (y, res) with
    ``res[:, :, b]`` the Jacobian of column b (n_in x n_in x B)."""
    n, B = xs.shape
    res = np.zeros((n, n, B), dtype=xs.dtype)
    y = None
    for i in range(n):
        seed = np.zeros_like(xs)
        seed[i, :] = 1
        if use_jvp:
            y, col = mlp_jvp(net, flat, xs, seed, ys)
            res[:, i, :] = col          # utils.jl:30-32
        else:
            y, row = mlp_vjp(net, flat, xs, seed, ys)
            res[i, :, :] = row          # utils.jl:12-13
    return y, res


# --------------------------------------------------------------------------------------
# the hot path: augmented_f (a1, a2, a3)
# --------------------------------------------------------------------------------------
def n_augment(train: bool) -> int:
    """src/icnf.jl:106-108 (TrainMode -> 2) / src/base_icnf.jl:79-81 (otherwise 0)."""
    return 2 if train else 0


def _colnorm(x):
    return np.sqrt(np.sum(x * x, axis=0))


def augmented_f_train(net: Net, flat, u, eps, norm_z: bool, norm_j: bool, use_jvp=False, ys=None):
    """Matrix/Train. VJP: src/icnf.jl:318-350; JVP: src/icnf.jl:384-420.
    u: D x B with D = n_in + 3; eps: n_in x B. Returns du: D x B."""
    n_aug = n_augment(True)
    z = u[: u.shape[0] - n_aug - 1, :]                       # icnf.jl:330
    if use_jvp:
        zdot, eJ = mlp_jvp(net, flat, z, eps, ys)            # icnf.jl:397-402
    else:
        zdot, eJ = mlp_vjp(net, flat, z, eps, ys)            # icnf.jl:331-332
    ldot = -np.sum(eJ * eps, axis=0, keepdims=True)          # icnf.jl:334 / :404
    Edot = _colnorm(zdot)[None, :] if norm_z else np.zeros_like(ldot)   # icnf.jl:335-341
    ndot = _colnorm(eJ)[None, :] if norm_j else np.zeros_like(ldot)     # icnf.jl:342-348
    return np.vstack([zdot, ldot, Edot, ndot])               # icnf.jl:349


def augmented_f_test(net: Net, flat, u, use_jvp=False, ys=None):
    """Matrix/Test exact trace: src/icnf.jl:148-164 with src/utils.jl:1-36."""
    z = u[: u.shape[0] - 1, :]
    zdot, J = jacobian_batched(net, flat, z, use_jvp, ys)
    ldot = -np.trace(J, axis1=0, axis2=1)[None, :]           # icnf.jl:162
    return np.vstack([zdot, ldot])                           # icnf.jl:163


# --------------------------------------------------------------------------------------
# model config mirror (only what the path needs; src/icnf.jl:69-104, base_icnf.jl:1-77)
# --------------------------------------------------------------------------------------
@dataclass
class Cfg:
    net: Net
    nvars: int
    naugs: int = 0
    lam1: float = 0.0
    lam2: float = 0.0
    lam3: float = 0.0
    use_jvp: bool = False
    tspan: tuple = (0.0, 1.0)

    @property
    def n_in(self):
        return self.nvars + self.naugs

    def D(self, train: bool):
        return self.n_in + 1 + n_augment(train)

    def rhs(self, flat, eps, train: bool, ys=None):
        """ys: conditioning inputs (n_cond x B) of the Cond* models (src/base_icnf.jl:288-309)."""
        if train:
            return lambda u: augmented_f_train(self.net, flat, u, eps, self.lam1 != 0,
                                               self.lam2 != 0, self.use_jvp, ys)
        return lambda u: augmented_f_test(self.net, flat, u, self.use_jvp, ys)


def inference_u0(cfg: Cfg, xs, train: bool):
    """src/base_icnf.jl:275-276,282: u0 = vcat(xs, zeros(naugs + n_aug + 1, B))."""
    zrs = np.zeros((cfg.naugs + n_augment(train) + 1, xs.shape[1]), dtype=xs.dtype)
    return np.vstack([xs, zrs])


# --------------------------------------------------------------------------------------
# Tsit5 (a7).  Third-party in the reference (OrdinaryDiffEq; base_icnf.jl:141); tableau
# from Tsitouras 2011, verified against the order conditions in tests/test_oracle.py.
# --------------------------------------------------------------------------------------
TSIT5_C = (0.0, 0.161, 0.327, 0.9, 0.9800255409045097, 1.0, 1.0)
TSIT5_A = (
    (),
    (0.161,),
    (-0.008480655492356989, 0.335480655492357),
    (2.8971530571054935, -6.359448489975075, 4.3622954328695815),
    (5.325864828439257, -11.748883564062828, 7.4955393428898365, -0.09249506636175525),
    (5.86145544294642, -12.92096931784711, 8.159367898576159, -0.071584973281401,
     -0.028269050394068383),
    (0.09646076681806523, 0.01, 0.4798896504144996, 1.379008574103742, -3.290069515436081,
     2.324710524099774),
)
TSIT5_B = TSIT5_A[6] + (0.0,)
TSIT5_BTILDE = (-0.00178001105222577714, -0.0008164344596567469, 0.007880878010261995,
                -0.1447110071732629, 0.5823571654525552, -0.45808210592918697,
                0.015151515151515152)


@dataclass
class SolveStats:
    nf: int = 0
    naccept: int = 0
    nreject: int = 0
    dts: list = field(default_factory=list)


def _rms(x):
    return math.sqrt(float(np.sum(np.square(x, dtype=np.float64))) / x.size)


def tsit5_step(f, u, k1, dt):
    """One Tsit5 step from (u, k1=f(u)). Returns (u_new, k7, err) with
    err = dt * sum(btilde_i k_i).  6 new RHS evaluations (FSAL)."""
    T = u.dtype.type
    ks = [k1]
    for s in range(1, 7):
        acc = T(TSIT5_A[s][0]) * ks[0]
        for j in range(1, s):
            acc = acc + T(TSIT5_A[s][j]) * ks[j]
        us = u + T(dt) * acc
        ks.append(f(us))
    u_new = us                               # a7 == b: stage-7 input is the new state
    e = T(TSIT5_BTILDE[0]) * ks[0]
    for j in range(1, 7):
        e = e + T(TSIT5_BTILDE[j]) * ks[j]
    return u_new, ks[6], T(dt) * e


def tsit5_solve(f, u0, t0, t1, *, dt=None, adaptive=True, abstol=1e-6, reltol=1e-3,
                maxiters=100000):
    """Tsit5 from t0 to t1 (either direction).  ``adaptive=False`` takes fixed steps of
    size ``dt`` (last one clipped).  Adaptive control follows OrdinaryDiffEq's published
    scheme as recalled in SURVEY.md Appendix A (Hairer initial dt, RMS norm over all D*B
    entries, PI controller beta1=7/50, beta2=2/25, gamma=0.9, qmin=0.2, qmax=10) --
    third-party, restated from memory, unverifiable here; strict parity uses fixed dt.
    Time and dt are kept in the state dtype, as the reference does (tspan::NTuple{2,T})."""
    T = u0.dtype.type
    t0, t1 = T(t0), T(t1)
    tdir = T(1) if t1 >= t0 else T(-1)
    st = SolveStats()
    u = u0.copy()
    k1 = f(u)
    st.nf += 1
    t = t0
    if adaptive and dt is None:
        # Hairer/OrdinaryDiffEq initial step (2 RHS evals; f0 doubles as k1)
        sk = T(abstol) + np.abs(u) * T(reltol)
        d0 = _rms(u / sk)
        d1 = _rms(k1 / sk)
        dt0 = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
        dt0 = T(min(dt0, abs(float(t1 - t0))))
        f1 = f(u + tdir * dt0 * k1)
        st.nf += 1
        d2 = _rms((f1 - k1) / sk) / float(dt0)
        m = max(d1, d2)
        dt1 = max(1e-6, float(dt0) * 1e-3) if m <= 1e-15 else (0.01 / m) ** (1.0 / 5.0)
        dt = T(min(100.0 * float(dt0), dt1, abs(float(t1 - t0))))
    dt = T(abs(dt))
    beta1, beta2, gamma, qmin, qmax = 7.0 / 50.0, 2.0 / 25.0, 0.9, 0.2, 10.0
    qold = 1e-4
    it = 0
    while it < maxiters:
        it += 1
        remaining = T(abs(t1 - t))
        h = dt if dt < remaining else remaining
        u_new, k7, err = tsit5_step(f, u, k1, tdir * h)
        st.nf += 6
        if not adaptive:
            accept, EEst = True, 0.0
        else:
            sc = T(abstol) + np.maximum(np.abs(u), np.abs(u_new)) * T(reltol)
            EEst = _rms(err / sc)
            accept = EEst <= 1.0
        if adaptive:
            q11 = max(EEst, 1e-30) ** beta1
            q = q11 / (qold ** beta2)
            q = max(1.0 / qmax, min(1.0 / qmin, q / gamma))
        if accept:
            st.naccept += 1
            st.dts.append(float(h))
            t = T(t + tdir * h)
            u, k1 = u_new, k7
            if adaptive:
                if 1.0 <= q <= 1.2:
                    q = 1.0
                qold = max(EEst, 1e-4)
                dt = T(float(h) / q) if h == dt else T(float(dt))  # clipped last step keeps dt
            if abs(float(t1 - t)) <= 100.0 * float(np.finfo(u.dtype).eps) * max(1.0, abs(float(t1))):
                return u, st
        else:
            st.nreject += 1
            dt = T(float(h) / min(1.0 / qmin, q11 / gamma))
    raise RuntimeError("maxiters reached")


# --------------------------------------------------------------------------------------
# post-processing + loss (a8, a9)
# --------------------------------------------------------------------------------------
def inference_sol(cfg: Cfg, fsol, train: bool):
    """src/base_icnf.jl:167-189.  Returns (logp_x [B], (Edot, ndot, Adot) rows [B each]);
    in TestMode the first two rows are absent in the reference; here they are returned
    as None."""
    n_aug = n_augment(train)
    D = fsol.shape[0]
    z = fsol[: D - n_aug - 1, :]
    dlogp = fsol[D - n_aug - 1, :]
    augs = fsol[D - n_aug:, :]
    n = z.shape[0]
    logpz = -0.5 * (n * math.log(2.0 * math.pi) + np.sum(z * z, axis=0))   # MvNormal(0, I)
    logpx = (logpz - dlogp).astype(fsol.dtype)
    if cfg.lam3 != 0 and cfg.naugs > 0:                                     # :179-182
        Adot = _colnorm(z[n - cfg.naugs:, :])
    else:
        Adot = np.zeros_like(dlogp)
    if train:
        return logpx, (augs[0], augs[1], Adot)
    return logpx, (None, None, Adot)


def loss(cfg: Cfg, logpx, regs, train: bool):
    """Train: src/icnf.jl:481-490; otherwise src/base_icnf.jl:489-497."""
    if train:
        E, n, A = regs
        return float(np.mean(-logpx + cfg.lam1 * E + cfg.lam2 * n + cfg.lam3 * A))
    return float(-np.mean(logpx))


def inference(cfg: Cfg, flat, xs, eps, train: bool, ys=None, **solve_kw):
    """src/base_icnf.jl:407-415: inference_prob -> solve -> inference_sol, with eps given
    (drawn once per call in the reference, base_icnf.jl:277-278)."""
    u0 = inference_u0(cfg, xs, train)
    fsol, st = tsit5_solve(cfg.rhs(flat, eps, train, ys), u0, cfg.tspan[0], cfg.tspan[1], **solve_kw)
    logpx, regs = inference_sol(cfg, fsol, train)
    return fsol, logpx, regs, st


# --------------------------------------------------------------------------------------
# BASELINE.json configs (SURVEY.md section 8 size table)
# --------------------------------------------------------------------------------------
def baseline_cfg(i: int) -> tuple:
    """Returns (Cfg, B, train) for BASELINE config i in 1..5."""
    T = (ACT_TANH,)
    if i == 1:
        return Cfg(Net((2, 6, 2), T * 2), 1, 1, 1e-2, 1e-2, 1e-2, tspan=(0.0, 13.0)), 1024, True
    if i == 2:
        return Cfg(Net((16, 48, 16), T * 2), 8, 8, 1e-2, 1e-2, 1e-2), 4096, True
    if i == 3:
        return Cfg(Net((32, 128, 128, 32), T * 3), 32, 0, 1e-2, 1e-2, 0.0), 8192, True
    if i == 4:
        return Cfg(Net((32, 128, 128, 32), T * 3), 32, 0, 0.0, 0.0, 0.0), 65536, True
    if i == 5:
        return Cfg(Net((128, 384, 128), T * 2), 64, 64, 1e-2, 1e-2, 1e-2), 2048, False
    raise ValueError(i)
