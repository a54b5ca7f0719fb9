"""The reference's README example (README.md:31-110) end to end on the device: RNODE 1+1,
Chain(Dense(2 => 6, tanh), Dense(6 => 2, tanh)), tspan (0, 13), steer 0.1, lambdas 1e-2, README
tolerances; 1024 draws of Beta(2, 4); ICNFModel defaults (Lion, 300 epochs, batch 32); then
pdf(ICNFDist(TestMode)) against the true density with the three distances the README prints and
test/regression_tests.jl:46-48 bounds by 0.1.

    python tools/readme_example.py [--epochs 300] [--n 1024] [--out gpurun_out/readme_example.json]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=300)
    ap.add_argument("--n", type=int, default=1024)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--eta", type=float, default=1e-3)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--naugs", type=int, default=1, help="README: naugs = nvars; 0 = the README's 'without augmentation' line")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    import continuousnf.jl_amd as cnf
    from continuousnf.jl_amd import mlj
    from scipy import stats

    nvars = 1
    naugs = a.naugs
    n_in = nvars + naugs
    nn = cnf.Chain(cnf.Dense(n_in, 3 * n_in, "tanh"), cnf.Dense(3 * n_in, n_in, "tanh"))
    eps32 = float(np.finfo(np.float32).eps)
    icnf = cnf.construct(cnf.RNODE, nn, nvars, naugs, compute_mode=cnf.HIPVecJacMatrixMode(),
                         tspan=(0.0, 13.0), steer_rate=0.1, lambda1=1e-2, lambda2=1e-2, lambda3=1e-2,
                         sol_kwargs=dict(reltol=float(np.sqrt(eps32)), abstol=eps32, maxiters=2**31 - 1),
                         rng=a.seed)
    rng = np.random.default_rng(a.seed)
    r = rng.beta(2.0, 4.0, size=(nvars, a.n)).astype(np.float32)

    seen = []

    def cb(it, val):
        seen.append(val)
        if it % 320 == 0:
            print(f"iteration {it}: mean loss of the last 320 batches {np.mean(seen[-320:]):.4f}", flush=True)
    model = mlj.ICNFModel(icnf, optimizers=(mlj.Lion(eta=a.eta),), n_epochs=a.epochs, batch_size=a.batch, callback=cb)
    t0 = time.perf_counter()
    fitresult, _, report = mlj.fit(model, 0, r.T)
    t_fit = time.perf_counter() - t0
    ps, st = fitresult
    d = cnf.ICNFDist(icnf, cnf.TestMode(), ps, st)
    actual_pdf = stats.beta(2.0, 4.0).pdf(r.reshape(-1))
    estimated_pdf = np.asarray(cnf.pdf(d, r)).reshape(-1)
    mad_ = float(np.mean(np.abs(estimated_pdf - actual_pdf)))                   # Distances.meanad
    msd_ = float(np.mean((estimated_pdf - actual_pdf) ** 2))                    # Distances.msd
    tv_dis = float(np.sum(np.abs(estimated_pdf - actual_pdf)) / 2 / a.n)        # totalvariation / n
    new_data = np.asarray(cnf.rand(d, a.n))
    res = dict(mad_=mad_, msd_=msd_, tv_dis=tv_dis, fit_seconds=t_fit, iterations=int(report["stats"]["iterations"]),
               first_loss=float(np.mean(report["losses"][:32])), last_loss=float(np.mean(report["losses"][-32:])),
               sample_mean=float(new_data.mean()), sample_var=float(new_data.var()),
               data_mean=float(r.mean()), data_var=float(r.var()), epochs=a.epochs, n=a.n, naugs=naugs)
    print(json.dumps(res))
    if a.out:
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1)
    ok = mad_ <= 0.1 and msd_ <= 0.1 and tv_dis <= 0.1                          # test/regression_tests.jl:46-48
    print("regression criteria (<= 0.1 each):", "PASS" if ok else "FAIL")
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
