"""Error of the wave-local gradient (k_solve_wave<GRAD>) and of the streamed gradient path against the float64 oracle on the
same accepted steps, and their timings, on the cases of test_loss_grad_wave_local_small_networks.
    python tools/wave_grad_check.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch
import continuousnf.jl_amd as cnf
from oracle import cnf_oracle as O, cnf_grad_oracle as G
import helpers

T2 = (O.ACT_TANH,) * 2
cases = [
    (O.Cfg(O.Net((16, 48, 16), T2), 8, 8, 1e-2, 1e-2, 1e-2, tspan=(0.0, 13.0)), 32, dict()),
    (O.Cfg(O.Net((2, 6, 2), T2), 1, 1, 1e-2, 1e-2, 1e-2, tspan=(0.0, 13.0)), 32, dict()),
    (O.Cfg(O.Net((16, 64, 16), T2), 10, 6, 0.0, 1e-2, 5e-2), 300, dict()),
    (O.Cfg(O.Net((16, 32, 16), T2), 16, 0, 0.0, 0.0, 0.0), 2048, dict()),
    (O.Cfg(O.Net((16, 48, 16), T2), 8, 8, 1e-2, 1e-2, 1e-2), 4096, dict()),                  # BASELINE config 2's batch
]
for ci, (cfg, B, sol_kw) in enumerate(cases):
    rng = np.random.default_rng(900 + ci)
    flat = O.glorot_params(cfg.net, rng, np.float32, float(os.environ.get("SCALE", "0.5")))
    xs = rng.standard_normal((cfg.nvars, B)).astype(np.float32)
    eps = rng.standard_normal((cfg.n_in, B)).astype(np.float32)
    dxs, deps = torch.from_numpy(xs).cuda(), torch.from_numpy(eps).cuda()
    for kernel in ("mfma", "generic"):
        ic = helpers.make_icnf(cnf, cfg, kernel=kernel, sol_kwargs=dict(sol_kw))
        val, grad, gx = cnf.loss_and_grad(ic, cnf.TrainMode(), dxs, flat, {}, eps=deps, with_x=True)
        st = dict(ic.last_stats)
        c64 = O.Cfg(cfg.net, cfg.nvars, cfg.naugs, cfg.lam1, cfg.lam2, cfg.lam3, tspan=cfg.tspan)
        rval, rgrad, ost = G.loss_and_grad(c64, flat.astype(np.float64), xs.astype(np.float64), eps.astype(np.float64), None,
                                           dts=[float(d) for d in ic.last_steps])
        g = grad.cpu().numpy(); gxn = gx.cpu().numpy()
        sc = np.abs(rgrad).max() + np.sqrt(np.mean(rgrad ** 2)); scx = np.abs(ost.grad_x).max() + np.sqrt(np.mean(ost.grad_x ** 2))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            cnf.loss_and_grad(ic, cnf.TrainMode(), dxs, flat, {}, eps=deps)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 20 * 1e3
        print(f"{cfg.net.dims} B={B} {kernel}: steps {st['naccept']}+{st['nreject']} launches {st['launches']} loss err {abs(val - rval):.2e} "
              f"grad err/scale {np.abs(g - rgrad).max() / sc:.2e} gx err/scale {np.abs(gxn - ost.grad_x).max() / scx:.2e}  {ms:.3f} ms per loss_and_grad", flush=True)
        ic.close()
