#!/usr/bin/env python3
"""Is the blow-up of the README's AUGMENTED example (naugs = nvars: the loss runs below the entropy of the data)
a property of the objective or a bug of the device gradient?  N optimiser steps (Lion, the ICNFModel default) from the
same initial parameters with the same mini-batches, probes eps and steered end times, once with the gradient of
cnf_loss_grad (HIP) and once with oracle/cnf_grad_oracle.py (float64, pinned by torch autograd); the two loss
trajectories are written side by side.  If they track each other the objective is what diverges.
Lives under tests/ because it uses the oracle.     python tests/train_trajectory_check.py [out.json] [steps] [eta]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")


def lion(ps, m, g, eta, b1=0.9, b2=0.999):
    ps -= eta * np.sign(b1 * m + (1 - b1) * g)
    m *= b2
    m += (1 - b2) * g


def run(steps=200, eta=1e-3, nvars=1, naugs=1, B=32, seed=1, tspan=(0.0, 13.0), steer=0.1):
    import torch

    import continuousnf.jl_amd as cnf
    from oracle import cnf_grad_oracle as G
    from oracle import cnf_oracle as O
    n_in = nvars + naugs
    net = O.Net((n_in, 3 * n_in, n_in), (O.ACT_TANH, O.ACT_TANH))
    nn = cnf.Chain(cnf.Dense(n_in, 3 * n_in, "tanh"), cnf.Dense(3 * n_in, n_in, "tanh"))
    e32 = float(np.finfo(np.float32).eps)
    kw = dict(reltol=float(np.sqrt(e32)), abstol=e32)
    rng = np.random.default_rng(seed)
    data = rng.beta(2.0, 4.0, size=(nvars, 1024)).astype(np.float32)
    flat0 = O.glorot_params(net, rng, np.float32, 0.0)
    # the draws both runs share
    batches = [rng.choice(1024, B, replace=False) for _ in range(steps)]
    epss = [rng.standard_normal((n_in, B)).astype(np.float32) for _ in range(steps)]
    t1s = [float(np.float32(tspan[1] + abs(tspan[1] - tspan[0]) * rng.uniform(-steer, steer))) for _ in range(steps)]
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
    out = {"hip": [], "oracle": [], "grad_rel_diff": [], "grad_rel_diff_replay": []}
    ps_h, m_h = flat0.astype(np.float64).copy(), np.zeros(flat0.size)
    ps_o, m_o = flat0.astype(np.float64).copy(), np.zeros(flat0.size)
    for k in range(steps):
        xs = data[:, batches[k]]
        icnf = cnf.construct(cnf.RNODE, nn, nvars, naugs, compute_mode=cnf.HIPVecJacMatrixMode(), tspan=(tspan[0], t1s[k]),
                             lambda1=1e-2, lambda2=1e-2, lambda3=1e-2, sol_kwargs=kw)
        val_h, g_h = cnf.loss_and_grad(icnf, cnf.TrainMode(), dev(xs), ps_h.astype(np.float32), {}, eps=dev(epss[k]))
        g_h = g_h.cpu().numpy().astype(np.float64)
        icnf.close()
        cfg = O.Cfg(net, nvars, naugs, 1e-2, 1e-2, 1e-2, tspan=(tspan[0], t1s[k]))
        val_o, g_o, _ = G.loss_and_grad(cfg, ps_o, xs.astype(np.float64), epss[k].astype(np.float64), None, **kw)
        # the gradients at the SAME point (the oracle's parameters), to separate gradient error from trajectory drift
        if k % 10 == 0:
            ic2 = cnf.construct(cnf.RNODE, nn, nvars, naugs, compute_mode=cnf.HIPVecJacMatrixMode(), tspan=(tspan[0], t1s[k]),
                                lambda1=1e-2, lambda2=1e-2, lambda3=1e-2, sol_kwargs=kw)
            _, g2 = cnf.loss_and_grad(ic2, cnf.TrainMode(), dev(xs), ps_o.astype(np.float32), {}, eps=dev(epss[k]))
            g2 = g2.cpu().numpy().astype(np.float64)
            # ... and against the oracle differentiating exactly the steps the device took (the same discrete map)
            _, g_r, _ = G.loss_and_grad(cfg, ps_o, xs.astype(np.float64), epss[k].astype(np.float64), None,
                                        dts=[float(d) for d in ic2.last_steps])
            ic2.close()
            out["grad_rel_diff"].append(float(np.abs(g2 - g_o).max() / (np.abs(g_o).max() + 1e-30)))
            out["grad_rel_diff_replay"].append(float(np.abs(g2 - g_r).max() / (np.abs(g_r).max() + 1e-30)))
        out["hip"].append(float(val_h)); out["oracle"].append(float(val_o))
        out.setdefault("param_diff", []).append(float(np.abs(ps_h - ps_o).max()))      # before this step's update
        lion(ps_h, m_h, g_h, eta); lion(ps_o, m_o, g_o, eta)
    out["max_abs_loss_diff"] = float(np.max(np.abs(np.array(out["hip"]) - np.array(out["oracle"]))))
    out["param_diff_final"] = float(np.abs(ps_h - ps_o).max())
    out["config"] = dict(steps=steps, eta=eta, nvars=nvars, naugs=naugs, B=B, tspan=tspan, steer=steer)
    return out


if __name__ == "__main__":
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "train_trajectory.json")
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    eta = float(sys.argv[3]) if len(sys.argv) > 3 else 1e-3
    nvars = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    res = run(steps, eta, nvars=nvars, naugs=nvars)
    json.dump(res, open(path, "w"), indent=1)
    h, o = np.array(res["hip"]), np.array(res["oracle"])
    for k in range(0, steps, max(1, steps // 20)):
        print(f"step {k:4d}  loss HIP {h[k]: .5f}  oracle {o[k]: .5f}")
    print("max |loss diff|", res["max_abs_loss_diff"], " max grad rel diff at equal parameters: own adaptive steps",
          max(res["grad_rel_diff"]), " the device's steps replayed", max(res["grad_rel_diff_replay"]),
          " final parameter diff", res["param_diff_final"])
