"""Runs every BASELINE.json configuration at its full size on cuda:0 and prints one JSON line
per configuration: which kernel ran, RHS evaluations per second of an adaptive solve (README
tolerances) and of a fixed-dt solve, plain-RHS launch time, and the parity error of the RHS
against the float32 C oracle on a column sample.  A checker script (it uses the oracle, so it lives under tests/); fills the
tables of DESIGN.md / README.md.    python tests/measure_configs.py [cfg ...]"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import continuousnf.jl_amd as cnf
from continuousnf.jl_amd import _lib
from oracle import c_oracle as CO
from oracle import cnf_oracle as O
from tests.helpers import make_icnf, parity_err

which = [int(a) for a in sys.argv[1:]] or [1, 2, 3, 4, 5, 13, 14, 23, 12, 15, 25]
dev = torch.device("cuda", 0)
for case in which:
    # 13 / 14: configs 3 / 4 at 65536 columns on ONE GPU (several tiles per workgroup of the one-launch solve);
    # 23: config 3's network in TestMode (exact trace of a three-layer net: k_trace3s, fused attempts)
    i = case % 10
    cfg, B, train = O.baseline_cfg(i)
    if case == 4:
        B = 8192                      # per-GPU shard of the 65536-column batch
    if case in (13, 14):
        B = 65536
    # beyond the BASELINE batches (round 5: no batch-size cliffs): config 2 at 32768 columns (k_solve_wave, four tiles per workgroup),
    # config 5 at 4096 / 16384 (k_solve_bcast, several tiles per workgroup)
    B = {12: 32768, 15: 4096, 25: 16384}.get(case, B)
    rng = np.random.default_rng(i)
    flat = O.glorot_params(cfg.net, rng, np.float32)
    modes = [("train", True)] if train else [("test", False), ("train", True)]
    if case in (3, 5):
        modes.append(("train-jvp", True))
    if case == 23:
        modes = [("test", False)]
    for mname, tr in modes:
        cfg.use_jvp = mname == "train-jvp"
        icnf = make_icnf(cnf, cfg, jvp=cfg.use_jvp, kernel="auto")
        icnf.set_params(flat)
        l, h = _lib.lib(), icnf.handle()
        D = cfg.D(tr)
        u_h = rng.standard_normal((D, B)).astype(np.float32)
        eps_h = rng.standard_normal((cfg.n_in, B)).astype(np.float32)
        u = torch.from_numpy(np.ascontiguousarray(u_h.T)).to(dev).reshape(-1)
        eps = torch.from_numpy(np.ascontiguousarray(eps_h.T)).to(dev).reshape(-1)
        du = torch.empty_like(u)
        sp = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        m = 1 if tr else 0
        kern = l.cnf_kernel_for(h, m, B)
        # plain RHS
        nrep = 20 if kern == 2 or i < 5 else 3
        _lib.check(l.cnf_rhs(h, m, 0, u.data_ptr(), eps.data_ptr(), du.data_ptr(), B, sp), h)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(nrep):
            _lib.check(l.cnf_rhs(h, m, 0, u.data_ptr(), eps.data_ptr(), du.data_ptr(), B, sp), h)
        torch.cuda.synchronize()
        rhs_us = (time.perf_counter() - t0) / nrep * 1e6
        # parity on a column sample
        ns = min(B, 128 if (i == 5 and not tr) else (256 if case in (15, 25) else 512))
        got = du.view(B, D)[:ns].cpu().numpy().T
        ref = CO.rhs(cfg, flat, u_h[:, :ns], eps_h[:, :ns], tr)
        perr = parity_err(got, ref, trace_row=cfg.n_in)
        fl, by = C.c_double(), C.c_double()
        l.cnf_rhs_work(h, m, B, C.byref(fl), C.byref(by))
        out = {"cfg": i, "case": case, "mode": mname, "B": B, "kernel": {1: "generic", 2: "mfma"}[kern], "rhs_us": round(rhs_us, 1),
               "rhs_TFLOPs": round(fl.value / rhs_us / 1e6, 2), "parity_err_vs_c_oracle": float(f"{perr:.2e}")}
        if not tr and len(cfg.net.dims) > 3:
            # (cnf_rhs_work prices the reference's n_in tangent sweeps for a >= 3-layer exact trace; the kernel executes the
            # re-associated form tr(D3 W3 D2 (W2 D1 W1)): forward + one d2 x d1 x d0 product per sample + the contraction)
            d = cfg.net.dims
            M = sum(a * b for a, b in zip(d[:-1], d[1:]))
            ex = B * (2.0 * M + 2.0 * d[0] * d[1] * d[2] + 2.0 * d[2] * d[3])
            out["rhs_TFLOPs_reference_formulation"] = out.pop("rhs_TFLOPs")
            out["rhs_TFLOPs"] = round(ex / rhs_us / 1e6, 2)
            out["rhs_flops_note"] = "executed (re-associated) formulation; split-bf16 products: ceiling 416.7 TFLOP/s fp32-equivalent"
        # solves
        u0 = u.clone()
        u0.view(B, D)[:, cfg.n_in:] = 0
        for tag, opts in (("adaptive", _lib.cnf_solve_opts(cfg.tspan[0], cfg.tspan[1], 1.1920929e-7, 3.4526698e-4, 0.0, 1, 1 << 20, 0)),
                          ("fixed", _lib.cnf_solve_opts(cfg.tspan[0], cfg.tspan[1], 0.0, 0.0, (cfg.tspan[1] - cfg.tspan[0]) / 32, 0, 1 << 20, 0))):
            if kern == 1 and i == 5 and not tr and tag == "fixed":
                continue
            if B > 8192 and tag == "fixed":
                continue
            stats = _lib.cnf_solve_stats()
            run = lambda: _lib.check(l.cnf_solve_tsit5(h, m, u0.data_ptr(), eps.data_ptr(), du.data_ptr(), B,
                                                       C.byref(opts), C.byref(stats), sp), h)
            run()
            n, reps = (5, 3) if kern == 2 else (1, 1)
            el = 1e9
            for _ in range(reps):          # best of `reps` batches: the host thread shares its cores
                t0 = time.perf_counter()
                for _ in range(n):
                    run()
                torch.cuda.synchronize()
                el = min(el, (time.perf_counter() - t0) / n)
            out[f"{tag}_ms"] = round(el * 1e3, 3)
            out[f"{tag}_nf"] = stats.nf
            out[f"{tag}_rhs_evals_per_s"] = round(stats.nf / el, 1)
            out[f"{tag}_launches"] = stats.launches
        print(json.dumps(out), flush=True)
        icnf.close()
