"""BLAS-backed CPU baseline of the hot path (bench.py's ``cpu_baseline`` leg; TEST INFRASTRUCTURE ONLY,
same rules as the rest of ``oracle/``).

The reference's CPU path evaluates ``augmented_f`` Matrix/Train/VJP (src/icnf.jl:318-350) as sgemm over the
whole ``n x B`` activation matrices -- Lux ``Dense`` forward and the Enzyme-generated reverse pass both land in
BLAS (src/icnf.jl:331-332; third party) -- plus broadcast ``tanh_fast``.  This file restates exactly that shape
of computation in float32 on torch-CPU (multi-threaded sgemm and vectorised tanh over all host cores), driven
by the oracle's own Tsit5 loop (``cnf_oracle.tsit5_solve``), so that the GPU number has a CPU number of the
same class beside it.  ``oracle/cnf_oracle.c`` (scalar loops over 16-sample blocks) stays as the second,
naive port.  PARITY UNPINNED against Julia, like the oracle it is checked against (tests/test_oracle.py)."""
from __future__ import annotations

import numpy as np

from . import cnf_oracle as O


def threads() -> int:
    import torch
    return torch.get_num_threads()


def available_cores() -> int:
    """Cores this process may really use: the affinity mask, capped by a cgroup CPU quota (a GPU box hands a
    container a share of the host's cores; torch would otherwise start one thread per host core)."""
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def tune_threads(cfg, flat, u0, eps, reps=3):
    """Pick the torch thread count that evaluates the RHS fastest (candidates around the available cores) and
    leave torch set to it.  Returns (threads, rhs_per_s)."""
    import time
    import torch
    avail = available_cores()
    cands = sorted({max(1, avail // 2), avail, min(2 * avail, os_cpu_count()), os_cpu_count()})
    f = make_rhs(cfg, flat, eps)
    best = (0.0, avail)
    for n in cands:
        torch.set_num_threads(n)
        f(u0)
        t0 = time.perf_counter()
        for _ in range(reps):
            f(u0)
        rate = reps / (time.perf_counter() - t0)
        if rate > best[0]:
            best = (rate, n)
    torch.set_num_threads(best[1])
    return best[1], best[0]


def os_cpu_count() -> int:
    import os
    return os.cpu_count() or 1


def make_rhs(cfg: O.Cfg, flat, eps):
    """Returns f(u) -> du for TrainMode/VJP on numpy float32 ``(D, B)`` arrays; all matrix work in torch-CPU."""
    import torch
    net = cfg.net
    if any(a not in (O.ACT_TANH, O.ACT_IDENTITY) for a in net.acts):
        raise NotImplementedError("baseline restates tanh / identity layers (the BASELINE workloads)")
    Ws, bs = O.unflatten_params(net, np.asarray(flat, dtype=np.float32))
    Ws = [torch.from_numpy(np.ascontiguousarray(W)) for W in Ws]
    WTs = [W.t().contiguous() for W in Ws]
    bs = [torch.from_numpy(np.ascontiguousarray(b)).reshape(-1, 1) for b in bs]
    ep = torch.from_numpy(np.ascontiguousarray(eps, dtype=np.float32))
    n_in, L = cfg.n_in, net.n_layers
    norm_z, norm_j = cfg.lam1 != 0, cfg.lam2 != 0

    def f(u):
        with torch.no_grad():
            z = torch.from_numpy(u)[:n_in]                      # slice (src/icnf.jl:330)
            hs, h = [], z
            for l in range(L):                                  # Dense forward: sgemm + bias + tanh
                a = torch.addmm(bs[l], Ws[l], h)
                h = torch.tanh(a) if net.acts[l] == O.ACT_TANH else a
                hs.append(h)
            zdot = h
            g = ep                                              # pullback of eps (src/icnf.jl:331-332)
            for l in range(L - 1, -1, -1):
                if net.acts[l] == O.ACT_TANH:
                    g = g * (1.0 - hs[l] * hs[l])
                g = WTs[l] @ g
            ldot = -(g * ep).sum(0, keepdim=True)               # src/icnf.jl:334
            zero = torch.zeros_like(ldot)
            E = torch.linalg.vector_norm(zdot, dim=0, keepdim=True) if norm_z else zero    # :335-341
            n = torch.linalg.vector_norm(g, dim=0, keepdim=True) if norm_j else zero       # :342-348
            return torch.cat([zdot, ldot, E, n]).numpy()        # vcat (src/icnf.jl:349)
    return f


def solve(cfg: O.Cfg, flat, u0, eps, **kw):
    """Adaptive / fixed Tsit5 solve of the TrainMode system; returns (u_final, stats dict).  numpy driver around the
    torch right-hand side (the round-2 baseline; kept for the cross-check in tests/test_oracle.py)."""
    f = make_rhs(cfg, flat, eps)
    u, st = O.tsit5_solve(f, np.ascontiguousarray(u0, dtype=np.float32), 0.0, 1.0, **kw)
    return u, {"nf": st.nf, "naccept": st.naccept, "nreject": st.nreject}


def make_rhs_torch(cfg: O.Cfg, flat, eps):
    """f(u) -> du on torch float32 ``(D, B)`` tensors, no numpy round trip: the same sgemm / tanh formulation as make_rhs,
    writing into a preallocated output."""
    import torch
    net = cfg.net
    if any(a not in (O.ACT_TANH, O.ACT_IDENTITY) for a in net.acts):
        raise NotImplementedError("baseline restates tanh / identity layers (the BASELINE workloads)")
    Ws, bs = O.unflatten_params(net, np.asarray(flat, dtype=np.float32))
    Ws = [torch.from_numpy(np.ascontiguousarray(W)) for W in Ws]
    WTs = [W.t().contiguous() for W in Ws]
    bs = [torch.from_numpy(np.ascontiguousarray(b)).reshape(-1, 1) for b in bs]
    ep = torch.from_numpy(np.ascontiguousarray(eps, dtype=np.float32))
    n_in, L = cfg.n_in, net.n_layers
    norm_z, norm_j = cfg.lam1 != 0, cfg.lam2 != 0

    def f(u):
        z = u[:n_in]
        hs, h = [], z
        for l in range(L):
            a = torch.addmm(bs[l], Ws[l], h)
            h = a.tanh_() if net.acts[l] == O.ACT_TANH else a
            hs.append(h)
        du = torch.empty_like(u)
        du[:n_in] = h
        g = ep
        for l in range(L - 1, -1, -1):
            if net.acts[l] == O.ACT_TANH:
                g = torch.addcmul(g, g * hs[l], hs[l], value=-1.0)        # g (1 - h^2)
            g = WTs[l] @ g
        du[n_in] = -(g * ep).sum(0)
        if norm_z: torch.linalg.vector_norm(h, dim=0, out=du[n_in + 1])
        else: du[n_in + 1].zero_()
        if norm_j: torch.linalg.vector_norm(g, dim=0, out=du[n_in + 2])
        else: du[n_in + 2].zero_()
        return du
    return f


def solve_torch(cfg: O.Cfg, flat, u0, eps, *, dt=None, adaptive=True, abstol=1e-6, reltol=1e-3, maxiters=100000):
    """The Tsit5 driver of cnf_oracle.tsit5_solve with every array operation (stage combinations, error estimate, norms)
    as a threaded torch-CPU op on float32 ``(D, B)`` tensors: what bench.py times as ``cpu_baseline``.  Same control law
    as the oracle, the C port and the GPU.  Returns (u_final numpy, stats dict)."""
    import math
    import torch
    a = _tsit5_tables()
    f = make_rhs_torch(cfg, flat, eps)
    with torch.no_grad():
        u = torch.from_numpy(np.ascontiguousarray(u0, dtype=np.float32)).clone()
        n = u.numel()
        t0, t1 = np.float32(0.0), np.float32(1.0)
        tdir = np.float32(1.0)
        rms = lambda x: math.sqrt(float(torch.dot(x.reshape(-1), x.reshape(-1))) / n)
        k = [None] * 7
        k[0] = f(u)
        nf, nacc, nrej = 1, 0, 0
        t = t0
        if adaptive and dt is None:
            sk = abstol + u.abs() * reltol
            d0, d1 = rms(u / sk), rms(k[0] / sk)
            dt0 = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
            dt0 = np.float32(min(dt0, abs(float(t1 - t0))))
            f1 = f(torch.add(u, k[0], alpha=float(tdir * dt0)))
            nf += 1
            d2 = rms((f1 - k[0]) / sk) / float(dt0)
            m = max(d1, d2)
            dt1 = max(1e-6, float(dt0) * 1e-3) if m <= 1e-15 else (0.01 / m) ** 0.2
            dt = np.float32(min(100.0 * float(dt0), dt1, abs(float(t1 - t0))))
        dt = np.float32(abs(dt))
        beta1, beta2, gamma, qmin, qmax = 7.0 / 50.0, 2.0 / 25.0, 0.9, 0.2, 10.0
        qold = 1e-4
        for _ in range(maxiters):
            rem = np.float32(abs(t1 - t))
            h = dt if dt < rem else rem
            hs = float(tdir * h)
            for s in range(1, 7):
                acc = k[0] * float(a["A"][s][0])
                for j in range(1, s):
                    acc.add_(k[j], alpha=float(a["A"][s][j]))
                us = torch.add(u, acc, alpha=hs)
                k[s] = f(us)
                nf += 1
            u_new = us                                   # a7 = b
            EEst, accept = 0.0, True
            if adaptive:
                err = k[0] * float(a["BT"][0])
                for j in range(1, 7):
                    err.add_(k[j], alpha=float(a["BT"][j]))
                sc = torch.maximum(u.abs(), u_new.abs()).mul_(reltol).add_(abstol)
                EEst = rms(err.mul_(hs).div_(sc))
                accept = EEst <= 1.0
                q11 = max(EEst, 1e-30) ** beta1
                q = q11 / (qold ** beta2)
                q = max(1.0 / qmax, min(1.0 / qmin, q / gamma))
            if accept:
                nacc += 1
                t = np.float32(t + tdir * h)
                u, k[0] = u_new, k[6]
                if adaptive:
                    if 1.0 <= q <= 1.2:
                        q = 1.0
                    qold = max(EEst, 1e-4)
                    dt = np.float32(float(h) / q) if h == dt else dt
                if abs(float(t1 - t)) <= 100.0 * float(np.finfo(np.float32).eps) * max(1.0, abs(float(t1))):
                    return u.numpy(), {"nf": nf, "naccept": nacc, "nreject": nrej}
            else:
                nrej += 1
                dt = np.float32(float(h) / min(1.0 / qmin, q11 / gamma))
    raise RuntimeError("maxiters reached")


def _tsit5_tables():
    A = [[], [0.161], [-0.008480655492356989, 0.335480655492357],
         [2.8971530571054935, -6.359448489975075, 4.3622954328695815],
         [5.325864828439257, -11.748883564062828, 7.4955393428898365, -0.09249506636175525],
         [5.86145544294642, -12.92096931784711, 8.159367898576159, -0.071584973281401, -0.028269050394068383],
         [0.09646076681806523, 0.01, 0.4798896504144996, 1.379008574103742, -3.290069515436081, 2.324710524099774]]
    BT = [-0.00178001105222577714, -0.0008164344596567469, 0.007880878010261995, -0.1447110071732629,
          0.5823571654525552, -0.45808210592918697, 0.015151515151515152]
    return {"A": A, "BT": BT}
