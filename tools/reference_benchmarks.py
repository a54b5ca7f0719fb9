"""The reference's own benchmark suite (benchmark/benchmarks.jl:24-99) on this implementation.

    RNODE, nvars = naugs = 8, nn = Chain(Dense(16 => 16, tanh)), n = 64 samples r = rand(Float32, 8, 64), tspan (0, 13),
    steer_rate 0.1, lambda3 = 1e-2 (lambda1 = lambda2 = 1e-2: RNODE defaults), solver defaults, ps = Lux.setup
    SUITE[main][no_inplace | inplace][direct][train | test]      = loss(icnf, TrainMode() | TestMode(), r, ps, st)
    SUITE[main][no_inplace | inplace][AD-1-order][train | test]  = gradient of that loss w.r.t. ps

(in-place and out-of-place are the same arithmetic behind the C ABI: one column each).  Reported like BenchmarkTools does:
minimum / median / mean time per call over `--samples` calls after a warm-up, in microseconds, with the launches per call;
beside each, the CPU restatement of the same call (oracle/: the C oracle for the losses, the numpy discrete adjoint for the
gradient -- the Julia package cannot run here, so there is no reference number) on a few calls.  The reference stores no
results for this suite (benchmark/*.json is git-ignored).

    python tools/reference_benchmarks.py [--samples 200] [--out gpurun_out/reference_benchmarks.json]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def timed(fn, samples, warm=10):
    import torch
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(samples):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(1e6 * (time.perf_counter() - t0))
    ts = np.asarray(ts)
    return dict(min_us=float(ts.min()), median_us=float(np.median(ts)), mean_us=float(ts.mean()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=200)
    ap.add_argument("--cpu-samples", type=int, default=3)
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    import torch
    import continuousnf.jl_amd as cnf
    from continuousnf.jl_amd import layers

    nvars = naugs = 8
    n_in, n = nvars + naugs, 64
    nn = cnf.Chain(cnf.Dense(n_in, n_in, "tanh"))
    icnf = cnf.construct(cnf.RNODE, nn, nvars, naugs, compute_mode=cnf.HIPVecJacMatrixMode(), tspan=(0.0, 13.0), steer_rate=0.1,
                         lambda3=1e-2, rng=1)
    ps, st = layers.setup(icnf.rng, nn, init="lux_v1")
    r = np.random.default_rng(1).random((nvars, n)).astype(np.float32)
    dr = torch.from_numpy(r).cuda()
    res = dict(suite="benchmark/benchmarks.jl:24-99", model="RNODE 8+8, Chain(Dense(16 => 16, tanh)), n = 64, tspan (0, 13), steer 0.1, lambda3 1e-2",
               samples=a.samples, device=torch.cuda.get_device_name(0), rows={})

    def row(name, fn, stats):
        fn()
        t = timed(fn, a.samples)
        t.update(launches=int(stats()["launches"]), nf=int(stats()["nf"]), naccept=int(stats()["naccept"]))
        res["rows"][name] = t
        print(f"{name:28s} min {t['min_us']:9.1f} us   median {t['median_us']:9.1f} us   mean {t['mean_us']:9.1f} us   "
              f"({t['launches']} launches, nf {t['nf']})", flush=True)

    row("direct/train", lambda: cnf.loss(icnf, cnf.TrainMode(), dr, ps, st), lambda: icnf.last_stats)
    row("direct/test", lambda: cnf.loss(icnf, cnf.TestMode(), dr, ps, st), lambda: icnf.last_stats)
    row("AD-1-order/train", lambda: cnf.loss_and_grad(icnf, cnf.TrainMode(), dr, ps, st), lambda: icnf.last_stats)
    row("AD-1-order/test", lambda: cnf.loss_and_grad(icnf, cnf.TestMode(), dr, ps, st), lambda: icnf.last_stats)

    # ---- the CPU restatement of the same calls (baseline only; steering off so that both sides solve the same span) ----
    from oracle import cnf_oracle as O, cnf_grad_oracle as G
    from oracle import c_oracle
    net = O.Net((n_in, n_in), (O.ACT_TANH,))
    cfg = O.Cfg(net, nvars, naugs, 1e-2, 1e-2, 1e-2, tspan=(0.0, 13.0))
    eps = np.random.default_rng(2).standard_normal((n_in, n)).astype(np.float32)
    tol = dict(reltol=1e-3, abstol=1e-6)                       # OrdinaryDiffEq's defaults (sol_kwargs is empty in the suite)
    cpu = {}
    def cpu_row(name, fn):
        fn()
        ts = []
        for _ in range(a.cpu_samples):
            t0 = time.perf_counter(); fn(); ts.append(1e6 * (time.perf_counter() - t0))
        cpu[name] = dict(median_us=float(np.median(ts)), impl="oracle/ (float32 C restatement, 1 thread)" if "direct" in name else "oracle/cnf_grad_oracle.py (numpy float64)")
        print(f"cpu {name:24s} median {cpu[name]['median_us']:12.1f} us   [{cpu[name]['impl']}]", flush=True)
    c_oracle.set_threads(1)                                    # 64 columns: one core (a thread team's barriers cost more than the work)
    def c_loss(train):
        u0 = O.inference_u0(cfg, r, train)
        fsol, _ = c_oracle.solve(cfg, ps, u0, eps if train else None, train, **tol)
        logpx, regs = c_oracle.post(cfg, fsol, train)
        return O.loss(cfg, logpx, regs, train)
    cpu_row("direct/train", lambda: c_loss(True))
    cpu_row("direct/test", lambda: c_loss(False))
    cpu_row("AD-1-order/train", lambda: G.loss_and_grad(cfg, ps.astype(np.float64), r.astype(np.float64), eps.astype(np.float64), **tol))
    cpu_row("AD-1-order/test", lambda: G.loss_and_grad_test(cfg, ps.astype(np.float64), r.astype(np.float64), **tol))
    res["cpu_restatement"] = cpu
    print(json.dumps(res))
    if a.out:
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)
    return 0


if __name__ == "__main__":
    sys.exit(main())
