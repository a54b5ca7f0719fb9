#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X backend (contract: see the task statement).

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): RNODE,
nvars=32, naugs=0, MLP 32->128->128->32 (tanh), batch 8192 per GPU, TrainMode
(Hutchinson VJP + regulariser rows), Tsit5 over tspan (0, 1) with the README tolerances
(reltol = sqrt(eps(Float32)), abstol = eps(Float32), README.md:64-65).

A "step" is one whole log-density inference (u0 assembly + adaptive Tsit5 solve + logpdf
post-processing + loss sums) over one synthetic Gaussian batch that is already resident in
HBM.  metric = RHS evaluations per second (the integrator's `nf` counter / wall time).
With N GPUs the batch axis is sharded (8192 columns per rank, weak scaling): every rank
solves its own shard, the only collective is the 5-float all-reduce behind the mean
log-likelihood, and `value` sums the shard-RHS-evaluations of all ranks.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_TFLOPS = 157.3     # MI355X_MICROARCH.md: fp32 MFMA = fp32 vector peak
PEAK_HBM_GBS = 8000.0       # spec


def cpu_baseline(B, flat, xs, eps, kw, budget_s=12.0):
    """The float32 C restatement of the reference path (oracle/cnf_oracle.c, OpenMP over
    all host cores) on the same workload; bounded to ~budget_s of CPU time.  The only place
    bench.py touches oracle/."""
    from oracle import c_oracle as CO
    from oracle import cnf_oracle as O
    cfg, _, _ = O.baseline_cfg(3)
    u0 = O.inference_u0(cfg, xs, True)
    CO.solve(cfg, flat, u0, eps, True, **kw)            # warm-up (page-in, thread pool)
    nf, t0, n = 0, time.perf_counter(), 0
    while True:
        _, st = CO.solve(cfg, flat, u0, eps, True, **kw)
        nf += st["nf"]
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= 50:
            break
    return {"value": nf / el, "unit": "RHS-evals/s", "cores": CO.threads(), "kind": "port",
            "sample": f"{n} adaptive Tsit5 solves of the same workload (B={B}, nf={st['nf']} each), "
                      f"C/OpenMP restatement of the reference path (the Julia package cannot run here)",
            "seconds": el}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--kernel", default="auto", choices=["auto", "generic", "mfma"])
    ap.add_argument("--batch", type=int, default=8192, help="columns per GPU")
    ap.add_argument("--fixed-dt", type=float, default=0.0, help="use fixed steps instead of adaptive")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    import continuousnf.jl_amd as cnf
    from continuousnf.jl_amd import _lib
    from continuousnf.jl_amd.parallel import allreduce_sums
    from continuousnf.jl_amd import configs

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # one rank per GPU; CNF_BENCH_BACKEND=gloo lets several ranks share one card for a rehearsal
    backend = os.environ.get("CNF_BENCH_BACKEND", "nccl")
    dev_index = local_rank % max(1, torch.cuda.device_count())
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    wl = configs.BASELINE[3]
    B = args.batch
    flat = configs.glorot_params(wl.dims, 12345)
    xs_h, eps_h = configs.synthetic_inputs(wl, B, 1 + rank)     # seeds 1..
    if args.fixed_dt > 0:
        kw = dict(adaptive=False, dt=args.fixed_dt)
    else:
        kw = dict(configs.README_TOLERANCES)
    icnf = configs.build(wl, kernel=args.kernel, sol_kwargs=kw)
    icnf.device = dev_index
    # resident in HBM in the reference's own layout (Julia column-major: a sample's rows contiguous)
    xs = torch.from_numpy(np.ascontiguousarray(xs_h.T)).to(dev).t()
    eps = torch.from_numpy(np.ascontiguousarray(eps_h.T)).to(dev).t()
    ps = torch.from_numpy(flat).to(dev)
    mode = cnf.TrainMode()

    def step():
        # one rank's share of `loss`: solve + post-processing + the 5 local sums in one C call, then the all-reduce
        _, _, local = cnf.inference(icnf, mode, xs, ps, {}, eps=eps, with_sums=True)
        sums = allreduce_sums(local)
        return icnf.last_stats, sums

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    nf_total = 0
    for _ in range(args.steps):
        st, sums = step()
        nf_total += st["nf"]
    sync()
    elapsed = time.perf_counter() - t0
    loss = cnf.loss_from_sums(icnf, mode, sums)
    t = torch.tensor([elapsed, float(nf_total)], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if world > 1:
        tmax = t.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed, nf_all = float(tmax[0]), float(tsum[1])
    else:
        nf_all = float(nf_total)

    # ---- roofline of the dominant kernel, measured live with HIP events on the launch stream
    roof = None
    if rank == 0:
        l, h = _lib.lib(), icnf.handle()
        import ctypes as C
        fl, by = C.c_double(), C.c_double()
        _lib.check(l.cnf_rhs_work(h, 1, B, C.byref(fl), C.byref(by)))
        stream = torch.cuda.current_stream(dev)
        sp = C.c_void_p(stream.cuda_stream)
        kernel_used = st["kernel_used"]
        if kernel_used == _lib.KERNEL_MFMA:
            # dominant kernel = the fused Tsit5 step kernel (6 RHS evaluations per launch):
            # time a fixed-dt solve of n steps = n launches of that kernel.
            nsteps = 256                  # long enough that the initial-dt phase and the finishing launch vanish in the mean
            opts = _lib.cnf_solve_opts(0.0, 1.0, 0.0, 0.0, 1.0 / nsteps, 0, 1 << 20, _lib.KERNEL_MFMA)
            stats = _lib.cnf_solve_stats()
            D = wl.n_in + 3
            u0 = torch.zeros(B * D, device=dev); u0.view(B, D)[:, :wl.nvars] = xs.t()
            out = torch.empty_like(u0)
            run = lambda: _lib.check(l.cnf_solve_tsit5(h, 1, u0.data_ptr(), eps.data_ptr(), out.data_ptr(), B,
                                                       C.byref(opts), C.byref(stats), sp), h)
            run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream); run(); e1.record(stream); e1.synchronize()
            per_launch_s = e0.elapsed_time(e1) * 1e-3 / nsteps
            units, kname = 6.0, "fused Tsit5 step kernel (6 RHS evaluations per launch)"
        else:
            D = wl.n_in + 3
            u = torch.randn(B * D, device=dev); du = torch.empty_like(u)
            n = 50
            for _ in range(5):
                _lib.check(l.cnf_rhs(h, 1, kernel_used, u.data_ptr(), eps.data_ptr(), du.data_ptr(), B, sp), h)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(n):
                _lib.check(l.cnf_rhs(h, 1, kernel_used, u.data_ptr(), eps.data_ptr(), du.data_ptr(), B, sp), h)
            e1.record(stream); e1.synchronize()
            per_launch_s = e0.elapsed_time(e1) * 1e-3 / n
            units, kname = 1.0, "k_rhs_generic (1 RHS evaluation per launch)"
        # HBM bytes per launch of that kernel from the PMC counters: collected in separate
        # rocprofv3 --pmc passes (tools/collect_profiles.sh; FETCH_SIZE and WRITE_SIZE cannot
        # share a pass) and committed under profiles/; bench.py cannot profile itself.
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "round1_step_kernel_pmc.json")
        if kernel_used == _lib.KERNEL_MFMA and os.path.exists(pmc) and B == 8192:
            traffic = json.load(open(pmc)).get("hbm_bytes_per_launch")
        tf = units * fl.value / per_launch_s / 1e12
        gbs = units * by.value / per_launch_s / 1e9
        roof = {"bound": "mfma", "achieved": tf, "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s",
                "frac": tf / PEAK_F32_TFLOPS, "traffic": traffic, "kernel": kname,
                "launch_us": per_launch_s * 1e6,
                "algorithmic_flops_per_launch": units * fl.value,
                "algorithmic_bytes_per_launch": units * by.value,
                "hbm_achieved_GBs": gbs, "hbm_frac": gbs / PEAK_HBM_GBS}

    if rank == 0:
        out = {
            "metric": "ODE RHS-evals/sec (nvars=32, batch=8192, Tsit5)",
            "value": nf_all / elapsed, "unit": "RHS-evals/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: RNODE nvars=32 naugs=0, MLP 32-128-128-32 tanh, "
                                   f"batch {B} per GPU, TrainMode Hutchinson VJP, Tsit5 tspan (0,1), "
                                   + (f"fixed dt={args.fixed_dt}" if args.fixed_dt > 0 else
                                      "adaptive reltol=sqrt(eps32) abstol=eps32"),
                       "global_batch": B * world, "parallelism": f"columns sharded x{world}",
                       "kernel": {1: "generic", 2: "mfma"}.get(st["kernel_used"], "?")},
            "sample_evals_per_s": nf_all / elapsed * B,
            "nf_per_solve": st["nf"], "naccept": st["naccept"], "nreject": st["nreject"],
            "launches_per_solve": st["launches"], "loss": loss,
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(B, flat, xs_h, eps_h,
                                               dict(dt=args.fixed_dt, adaptive=False) if args.fixed_dt > 0 else kw)
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
