"""CPU laboratory for the question DESIGN 7.0 left open: does the reference's training objective
(src/icnf.jl:481-490 over src/base_icnf.jl:266-286, 167-189), optimised the way src/exts/mlj_ext/core_icnf.jl:31-94 does,
reach a KNOWN optimum -- and if a run does not, which ingredient decides it?

Deliberately independent of the product and of oracle/: float64 torch, autograd for the gradient (no hand-written
adjoint), classical RK4 with a fixed number of steps instead of Tsit5 (another discretisation of the same ODE), the
two-layer field's trace terms written out in closed form.  What it shares with the product is only the reading of the
reference's objective, which is what is under test.  Nothing here runs on the GPU and nothing here is imported by the
product or the tests.

    python tools/train_lab.py --nvars 8 --naugs 0 --opt lion --eta 1e-3 --epochs 300 [--out file.json]

Optimisers: `lion` = Chen et al. 2023 (update from the OLD momentum: sign(b1 m + (1-b1) g), then m = b2 m + (1-b2) g);
`lion_opt` = the rule as Optimisers.jl states it, from memory (state = b2 g + (1-b2) state, step sign((b2-b1) g + b1 state):
with b2 = 0.999 the state IS the current gradient, i.e. sign-SGD); `signsgd`; `adam`.  `--decay` multiplies eta by a cosine
factor over the run (not in the reference: to separate "noise floor of a constant sign step" from "cannot get there")."""
import argparse
import json
import math
import sys
import time

import numpy as np
import torch

torch.set_num_threads(1)
DT = torch.float64
ACT = "tanh"


def init_params(rng, dims, kind):
    Ws, bs = [], []
    for i, o in zip(dims[:-1], dims[1:]):
        if kind == "glorot":
            lim = math.sqrt(6.0 / (i + o))
            W = rng.uniform(-lim, lim, size=(o, i)); b = np.zeros(o)
        elif kind == "lux":                     # PyTorch's Linear default: U(+-1/sqrt(in)) for both
            lim = 1.0 / math.sqrt(i)
            W = rng.uniform(-lim, lim, size=(o, i)); b = rng.uniform(-lim, lim, size=o)
        else:                                   # "lux_v1": Lux >= 1.0 Dense default as remembered: kaiming_uniform(gain(act)) with gain(tanh) = 5/3,
            gain = 5.0 / 3.0 if ACT == "tanh" else 1.0      # bound gain sqrt(3 / in); bias U(+-1/sqrt(in))
            lim = gain * math.sqrt(3.0 / i)
            W = rng.uniform(-lim, lim, size=(o, i)); b = rng.uniform(-1.0 / math.sqrt(i), 1.0 / math.sqrt(i), size=o)
        Ws.append(torch.tensor(W, dtype=DT, requires_grad=True)); bs.append(torch.tensor(b, dtype=DT, requires_grad=True))
    return Ws, bs


def make_rhs(Ws, bs, act, eps, lam1, lam2, exact):
    """augmented_f for a two-layer field, src/icnf.jl:318-350 (Train) / :148-164 (Test): rows [z; dlogp; E; n]."""
    W1, W2 = Ws; b1, b2 = bs
    n_in = W1.shape[1]
    C = (W1 * W2.T) if exact else None          # tr(D2 W2 D1 W1) = d1^T (W1 o W2^T) d2

    def f(u):
        z = u[:n_in]
        a1 = W1 @ z + b1[:, None]
        if act == "tanh":
            h1 = torch.tanh(a1); d1 = 1 - h1 * h1
            a2 = W2 @ h1 + b2[:, None]
            zd = torch.tanh(a2); d2 = 1 - zd * zd
        else:                                    # identity
            h1 = a1; d1 = torch.ones_like(a1)
            zd = W2 @ h1 + b2[:, None]; d2 = torch.ones_like(zd)
        if exact:
            ld = -((C @ d2) * d1).sum(0, keepdim=True)
            return torch.cat([zd, ld], 0)
        eJ = W1.T @ (d1 * (W2.T @ (d2 * eps)))   # eps^T J
        ld = -(eJ * eps).sum(0, keepdim=True)
        Ed = zd.norm(dim=0, keepdim=True) if lam1 else torch.zeros_like(ld)
        nd = eJ.norm(dim=0, keepdim=True) if lam2 else torch.zeros_like(ld)
        return torch.cat([zd, ld, Ed, nd], 0)
    return f


def solve_rk4(f, u, t1, nsteps):
    h = t1 / nsteps
    for _ in range(nsteps):
        k1 = f(u); k2 = f(u + 0.5 * h * k1); k3 = f(u + 0.5 * h * k2); k4 = f(u + h * k3)
        u = u + (h / 6) * (k1 + 2 * k2 + 2 * k3 + k4)
    return u


def loss_fn(Ws, bs, a, xs, rng, train=True):
    B = xs.shape[1]
    n_in = a.nvars + a.naugs
    exact = not train
    eps = torch.tensor(rng.standard_normal((n_in, B)), dtype=DT) if train else None
    t1 = a.T
    if train and a.steer:
        t1 = a.T + a.T * rng.uniform(-a.steer, a.steer)      # src/base_icnf.jl:108-121
    nrow = n_in + (3 if train else 1)
    u0 = torch.zeros(nrow, B, dtype=DT); u0[:a.nvars] = xs
    f = make_rhs(Ws, bs, a.act, eps, a.lam1 if train else 0, a.lam2 if train else 0, exact)
    u = solve_rk4(f, u0, t1, a.nsteps)
    z = u[:n_in]
    logpz = -0.5 * (n_in * math.log(2 * math.pi) + (z * z).sum(0))
    logpx = logpz - u[n_in]
    if not train:
        return logpx
    A = z[a.nvars:].norm(dim=0) if (a.lam3 and a.naugs) else torch.zeros(B, dtype=DT)
    return (-logpx + a.lam1 * u[n_in + 1] + a.lam2 * u[n_in + 2] + a.lam3 * A).mean(), (-logpx).mean()


def data_and_truth(a, rng):
    from scipy import stats
    if a.data == "beta":
        r = rng.beta(2.0, 4.0, size=(a.nvars, a.n))
        logp = stats.beta(2.0, 4.0).logpdf(r).sum(0)
        ent = a.nvars * float(stats.beta(2.0, 4.0).entropy())
    else:                                        # gauss: N(mu, sigma^2) per coordinate
        r = a.mu + a.sigma * rng.standard_normal((a.nvars, a.n))
        logp = stats.norm(a.mu, a.sigma).logpdf(r).sum(0)
        ent = a.nvars * float(stats.norm(a.mu, a.sigma).entropy())
    return r, logp, ent


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nvars", type=int, default=8); ap.add_argument("--naugs", type=int, default=0)
    ap.add_argument("--mult", type=int, default=3); ap.add_argument("--act", default="tanh")
    ap.add_argument("--n", type=int, default=1024); ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--epochs", type=int, default=300); ap.add_argument("--T", type=float, default=13.0)
    ap.add_argument("--steer", type=float, default=0.1); ap.add_argument("--nsteps", type=int, default=26)
    ap.add_argument("--lam1", type=float, default=1e-2); ap.add_argument("--lam2", type=float, default=1e-2)
    ap.add_argument("--lam3", type=float, default=1e-2)
    ap.add_argument("--opt", default="lion"); ap.add_argument("--eta", type=float, default=1e-3)
    ap.add_argument("--decay", action="store_true"); ap.add_argument("--init", default="glorot")
    ap.add_argument("--data", default="beta"); ap.add_argument("--mu", type=float, default=0.5)
    ap.add_argument("--sigma", type=float, default=0.25)
    ap.add_argument("--seed", type=int, default=1); ap.add_argument("--out", default="")
    a = ap.parse_args()
    global ACT
    ACT = a.act
    rng = np.random.default_rng(a.seed)
    n_in = a.nvars + a.naugs
    dims = (n_in, a.mult * n_in, n_in)
    Ws, bs = init_params(rng, dims, a.init)
    params = Ws + bs
    r, true_logp, ent = data_and_truth(a, rng)
    x = torch.tensor(r, dtype=DT)
    state = [torch.zeros_like(p) for p in params]; state2 = [torch.zeros_like(p) for p in params]
    b1, b2 = 0.9, 0.999
    nb = (a.n + a.batch - 1) // a.batch
    total = a.epochs * nb
    it = 0; marks = []; recent = []
    t0 = time.time()
    for ep in range(a.epochs):
        perm = rng.permutation(a.n)
        for lo in range(0, a.n, a.batch):
            xb = x[:, perm[lo:lo + a.batch]]
            L, nll = loss_fn(Ws, bs, a, xb, rng, True)
            gs = torch.autograd.grad(L, params)
            eta = a.eta * (0.5 * (1 + math.cos(math.pi * it / total)) if a.decay else 1.0)
            it += 1
            with torch.no_grad():
                for p, g, m, v in zip(params, gs, state, state2):
                    if a.opt == "lion":
                        p -= eta * torch.sign(b1 * m + (1 - b1) * g); m.mul_(b2).add_(g, alpha=1 - b2)
                    elif a.opt == "lion_opt":
                        m.mul_(1 - b2).add_(g, alpha=b2); p -= eta * torch.sign((b2 - b1) * g + b1 * m)
                    elif a.opt == "signsgd":
                        p -= eta * torch.sign(g)
                    elif a.opt == "adam":
                        m.mul_(b1).add_(g, alpha=1 - b1); v.mul_(b2).addcmul_(g, g, value=1 - b2)
                        p -= eta * (m / (1 - b1 ** it)) / ((v / (1 - b2 ** it)).sqrt() + 1e-8)
                    else:
                        raise SystemExit("unknown optimiser")
            recent.append(float(nll))
            if it % max(1, total // 20) == 0:
                marks.append((it, float(np.mean(recent[-nb:]))))
                print(f"it {it}/{total} mean batch NLL {marks[-1][1]:.4f}  (bound {ent:.4f})  {time.time() - t0:.0f}s", flush=True)
    with torch.no_grad():
        est_logp = loss_fn(Ws, bs, a, x, rng, False).numpy()
    est, act = np.exp(est_logp), np.exp(true_logp)
    res = dict(args=vars(a), iterations=it, entropy_bound=ent, test_nll_exact_trace=float(-est_logp.mean()),
               true_nll_on_sample=float(-true_logp.mean()),
               mad=float(np.mean(np.abs(est - act))), msd=float(np.mean((est - act) ** 2)),
               tv=float(np.sum(np.abs(est - act)) / 2 / a.n), true_pdf_mean=float(act.mean()), est_pdf_mean=float(est.mean()),
               mean_abs_logpdf_err=float(np.mean(np.abs(est_logp - true_logp))), marks=marks, seconds=time.time() - t0)
    print(json.dumps(res))
    if a.out:
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    sys.exit(main())
