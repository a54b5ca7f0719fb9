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
    """Adaptive / fixed Tsit5 solve of the TrainMode system; returns (u_final, stats dict)."""
    f = make_rhs(cfg, flat, eps)
    u, st = O.tsit5_solve(f, np.ascontiguousarray(u0, dtype=np.float32), 0.0, 1.0, **kw)
    return u, {"nf": st.nf, "naccept": st.naccept, "nreject": st.nreject}
