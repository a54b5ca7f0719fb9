#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X backend (contract: see the task statement).

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): RNODE,
nvars=32, naugs=0, MLP 32->128->128->32 (tanh), batch 8192 per GPU, TrainMode
(Hutchinson VJP + regulariser rows), Tsit5 over tspan (0, 1) with the README tolerances
(reltol = sqrt(eps(Float32)), abstol = eps(Float32), README.md:64-65).

A "step" is one whole log-density inference (u0 assembly + adaptive Tsit5 solve + logpdf
post-processing + loss sums) over one synthetic Gaussian batch that is already resident in
HBM.  metric = RHS evaluations per second (the integrator's `nf` counter / wall time).

N GPUs: one process per GPU, the batch axis sharded (8192 columns per rank, weak scaling).
Every rank solves its own shard; the only collective is the 5-float all-reduce behind the mean
log-likelihood (RCCL over xGMI, through the C ABI's cnf_loss_allreduce), and `value` sums the
shard-RHS-evaluations of all ranks.  Two ways in:
  * under a launcher (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`):
    RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the environment;
  * bare `python bench.py --gpus N`: this process starts the N rank processes itself, BEFORE it
    touches the GPU (it never re-executes a process that has initialised HIP), and waits for them.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")     # kernel arguments in device memory (see continuousnf.jl_amd/__init__.py)

PEAK_F32_TFLOPS = 157.3     # MI355X_MICROARCH.md: fp32 MFMA = fp32 vector peak
PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA
PEAK_HBM_GBS = 8000.0       # spec
PMC_FILE = os.path.join("profiles", "round4_step_kernel_pmc.json")   # tools/collect_profiles.sh + summarize_profiles.py


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--kernel", default="auto", choices=["auto", "generic", "mfma"])
    ap.add_argument("--batch", type=int, default=8192, help="columns per GPU")
    ap.add_argument("--fixed-dt", type=float, default=0.0, help="use fixed steps instead of adaptive")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-reference-suite", action="store_true", help="skip the reference's own benchmark suite (a second of small calls after the timed region)")
    ap.add_argument("--no-pmc", action="store_true", help="do not start the rocprofv3 child passes that measure roofline.traffic")
    ap.add_argument("--prewarm", type=float, default=0.4,
                    help="seconds of untimed solves before the W counted warm-up steps: holds the clocks (DVFS) so that a short "
                         "timed region does not sit on the ramp")
    ap.add_argument("--fp32-child", action="store_true", help=argparse.SUPPRESS)   # internal: the CNF_STEP_FP32=1 leg
    ap.add_argument("--depth", type=int, default=2, choices=[1, 2, 3],
                    help="steps in flight: 2 = each step is submitted while the one before it runs (default); 1 = one at a time")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start N rank processes (one per device) and wait.
    Nothing in this process has touched the GPU: torch.cuda.device_count() does not initialise it."""
    import torch
    backend = os.environ.get("CNF_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and args.gpus > ndev:
        sys.exit(f"bench.py: --gpus {args.gpus} but only {ndev} device(s) visible (one rank per GPU; "
                 f"CNF_BENCH_BACKEND=gloo rehearses several ranks on one card)")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        rc = max(rc, abs(p.wait()))
    sys.exit(rc)


def _timed_solves(fn, budget_s, max_n=50):
    fn()                                              # warm-up (page-in, thread pools)
    nf, n, t0 = 0, 0, time.perf_counter()
    while True:
        st = fn()
        nf += st["nf"]
        n += 1
        el = time.perf_counter() - t0
        if el > budget_s or n >= max_n:
            return nf, n, el, st


def cpu_baseline(B, flat, xs, eps, kw, budget_s=9.0):
    """CPU numbers beside the GPU number (the only place bench.py touches oracle/), both on a bounded
    sample of the same workload and both restatements -- the Julia package itself cannot run here:
      * float32 sgemm over the whole n x B matrices + vectorised tanh and the Tsit5 driver as threaded torch-CPU
        ops (oracle/cnf_blas.py) -- the class of computation the reference's Lux/BLAS path performs
        (src/icnf.jl:331-332);
      * the C/OpenMP restatement (oracle/cnf_oracle.c: register-blocked loops over 16-sample blocks), the second
        checker of the parity tests.
    Returned as (cpu_baseline, cpu_other): the FASTER of the two is the baseline, the other is reported beside it."""
    from oracle import cnf_blas as BL
    os.environ.setdefault("OMP_NUM_THREADS", str(BL.available_cores()))     # before the OpenMP port is loaded
    from oracle import c_oracle as CO
    from oracle import cnf_oracle as O
    march = CO.use_native()                               # -march=native on an AVX-512 host, else the portable x86-64-v3 build
    cfg, _, _ = O.baseline_cfg(3)
    u0 = O.inference_u0(cfg, xs, True)
    nthr, _ = BL.tune_threads(cfg, flat, u0, eps)         # the fastest thread count on this host (stated as `cores`)
    nf, n, el, st = _timed_solves(lambda: BL.solve_torch(cfg, flat, u0, eps, **kw)[1], budget_s)
    M = sum(a * b for a, b in zip(cfg.net.dims[:-1], cfg.net.dims[1:]))
    blas = {"value": nf / el, "unit": "RHS-evals/s", "cores": nthr, "host_cores": BL.os_cpu_count(),
            "available_cores": BL.available_cores(), "kind": "port",
            "achieved_gflops": nf / el * B * (4.0 * M + 6.0 * cfg.n_in) / 1e9,
            "nf_per_solve": st["nf"], "naccept": st.get("naccept"), "nreject": st.get("nreject"),
            "impl": "float32 sgemm over the full n x B matrices + tanh AND the Tsit5 driver (stage combinations, error "
                    "estimate, norms) as threaded torch-CPU ops on (D, B) tensors, no numpy round trips "
                    "(oracle/cnf_blas.py: solve_torch)",
            "sample": f"{n} adaptive Tsit5 solves of the same workload (B={B}, nf={st['nf']} each); restatement "
                      f"of the reference path (the Julia package cannot run here)",
            "seconds": el}
    CO.set_threads(BL.available_cores())                  # (torch may have sized the shared OpenMP runtime for the whole host)
    nf, n, el, st = _timed_solves(lambda: CO.solve(cfg, flat, u0, eps, True, **kw)[1], budget_s)
    port = {"value": nf / el, "unit": "RHS-evals/s", "cores": CO.threads(), "kind": "port",
            "host_cores": BL.os_cpu_count(), "available_cores": BL.available_cores(),
            "achieved_gflops": nf / el * B * (4.0 * M + 6.0 * cfg.n_in) / 1e9,
            "nf_per_solve": st["nf"], "naccept": st.get("naccept"), "nreject": st.get("nreject"),
            "impl": "C/OpenMP restatement: 16-sample blocks, four output rows per pass over the operand (gcc -O3 "
                    f"{march}, built on this host), libm tanhf (oracle/cnf_oracle.c)",
            "sample": f"{n} adaptive Tsit5 solves of the same workload (B={B}, nf={st['nf']} each); restatement "
                      f"of the reference path (the Julia package cannot run here)", "seconds": el}
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    blas["cpu_model"] = port["cpu_model"] = model
    return (blas, port) if blas["value"] >= port["value"] else (port, blas)


def measure_traffic_live(kernel_substr, timeout_s=150):
    """HBM bytes per launch of the headline kernel, measured NOW: two child runs of rocprofv3 (--pmc FETCH_SIZE, then
    --pmc WRITE_SIZE: they cannot share a pass; each with --kernel-trace only) over tools/prof_rhs.py bench -- adaptive
    solves of exactly this file's inputs --, combined as MI355X_MICROARCH.md's HBM section prescribes for gfx950
    (2 x FETCH_SIZE KiB + WRITE_SIZE KiB).  bench.py cannot profile itself; the children are ordinary processes started
    after the timed region.  Returns (bytes per launch, launches averaged) or (None, reason)."""
    import csv, glob, shutil, tempfile
    exe = shutil.which("rocprofv3")
    if not exe:
        return None, "rocprofv3 not on PATH"
    # never start a profiler from under a profiler: the outer tool's preloaded library has initialised the GPU in this
    # process tree, and the child launcher (a python script that then execs) is the forbidden exec on this pool
    if any(k in os.environ for k in ("ROCP_TOOL_LIBRARIES", "ROCPROFILER_LIBRARY_CTOR", "ROCPROF_OUTPUT_PATH", "ROCPROFILER_REGISTER_FORCE_LOAD")) \
            or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "already under a profiler"
    tmp = tempfile.mkdtemp(prefix="cnf_pmc_", dir="/tmp")
    env = {k: v for k, v in os.environ.items() if not (k.startswith("ROCP") or k == "LD_PRELOAD")}
    env.update(TMPDIR="/tmp", HIP_FORCE_DEV_KERNARG="1")
    vals = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [exe, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "--",
                   sys.executable, os.path.join(ROOT, "tools", "prof_rhs.py"), "bench", "8"]
            r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout_s)
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return None, f"rocprofv3 --pmc {counter} failed (rc {r.returncode})"
            v = [float(row["Counter_Value"]) for row in csv.DictReader(open(files[0]))
                 if kernel_substr in row["Kernel_Name"] and row["Counter_Name"] == counter]
            if not v:
                return None, f"no {counter} rows for {kernel_substr}"
            vals[counter] = (sum(v) / len(v), len(v))
    except Exception as e:                                     # (a timeout, a parse error: the committed figure is used instead)
        return None, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return 2.0 * vals["FETCH_SIZE"][0] * 1024.0 + vals["WRITE_SIZE"][0] * 1024.0, min(vals["FETCH_SIZE"][1], vals["WRITE_SIZE"][1])


def reference_suite(samples=100):
    """The four benchmarks of the reference's OWN suite (benchmark/benchmarks.jl:24-99: RNODE 8 + 8, Chain(Dense(16 => 16, tanh)),
    64 samples, tspan (0, 13), steer 0.1, lambda3 1e-2; loss in Train / TestMode and the gradient of each w.r.t. ps), outside
    the timed region, median microseconds per call (tools/reference_benchmarks.py prints the full table with the CPU
    restatement beside it).  The reference stores no results for this suite, so there is nothing to divide by."""
    import time
    import numpy as np
    import torch
    import continuousnf.jl_amd as cnf
    from continuousnf.jl_amd import layers
    nn = cnf.Chain(cnf.Dense(16, 16, "tanh"))
    ic = cnf.construct(cnf.RNODE, nn, 8, 8, compute_mode=cnf.HIPVecJacMatrixMode(), tspan=(0.0, 13.0), steer_rate=0.1, lambda3=1e-2, rng=1)
    ps, st = layers.setup(ic.rng, nn, init="lux_v1")
    dr = torch.from_numpy(np.random.default_rng(1).random((8, 64)).astype(np.float32)).cuda()
    rows = {"direct/train": lambda x: cnf.loss(ic, cnf.TrainMode(), x, ps, st), "direct/test": lambda x: cnf.loss(ic, cnf.TestMode(), x, ps, st),
            "AD-1-order/train": lambda x: cnf.loss_and_grad(ic, cnf.TrainMode(), x, ps, st),
            "AD-1-order/test": lambda x: cnf.loss_and_grad(ic, cnf.TestMode(), x, ps, st)}
    res = {"suite": "benchmark/benchmarks.jl:24-99", "unit": "us per call (median)", "samples": samples,
           "note": "median_us: the SAME data tensor every call, as the reference's suite does (the host mirror keeps the column-major copy "
                   "of the last data tensor); median_us_fresh_batch: a new nvars x n tensor per call, i.e. with the layout conversion a "
                   "mini-batch loop pays"}
    try:
        fresh = [dr.clone() for _ in range(samples)]          # (new tensor objects: the layout cache misses on each)
        for name, fn in rows.items():
            for _ in range(10):
                fn(dr)
            torch.cuda.synchronize()
            ts, tf = [], []
            for _ in range(samples):
                t0 = time.perf_counter(); fn(dr); torch.cuda.synchronize(); ts.append(1e6 * (time.perf_counter() - t0))
            for x in fresh:
                t0 = time.perf_counter(); fn(x); torch.cuda.synchronize(); tf.append(1e6 * (time.perf_counter() - t0))
            res[name] = {"median_us": float(np.median(ts)), "median_us_fresh_batch": float(np.median(tf)),
                         "launches": int(ic.last_stats["launches"]), "nf": int(ic.last_stats["nf"])}
    except Exception as e:            # (the headline line must not depend on this leg)
        res["error"] = repr(e)
    ic.close()
    return res


def gradient_leg(reps=5):
    """loss_and_grad (row f3, src/exts/mlj_ext/core_icnf.jl:59-73) on the headline network beside the forward loss, outside the
    timed region: the reference's training batch (32) and the BASELINE batch; milliseconds per call (tools/prof_grad.py prints
    the table over more shapes).  Informational: the metric of this bench is the forward path's."""
    import time
    import torch
    import continuousnf.jl_amd as cnf
    from continuousnf.jl_amd import configs
    res = {"unit": "ms per call", "network": "32-128-128-32 (config 3), adaptive Tsit5 at the README tolerances", "reps": reps}
    try:
        wl = configs.BASELINE[3]
        for B in (32, 8192):
            flat = torch.from_numpy(configs.glorot_params(wl.dims, 3, 0.05)).cuda()
            xs_h, eps_h = configs.synthetic_inputs(wl, B, 3)
            xs, eps = torch.from_numpy(xs_h).cuda(), torch.from_numpy(eps_h).cuda()
            ic = configs.build(wl, sol_kwargs=configs.README_TOLERANCES)
            row = {}
            for name, fn in (("loss", lambda: cnf.loss(ic, cnf.TrainMode(), xs, flat, {}, eps=eps)),
                             ("loss_and_grad", lambda: cnf.loss_and_grad(ic, cnf.TrainMode(), xs, flat, {}, eps=eps))):
                fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    fn()
                torch.cuda.synchronize()
                row[name] = round((time.perf_counter() - t0) / reps * 1e3, 3)
            row["accepted_steps"] = int(ic.last_stats["naccept"])
            res[f"B={B}"] = row
            ic.close()
    except Exception as e:            # (the headline line must not depend on this leg)
        res["error"] = repr(e)
    return res


def fp32_child_leg(args):
    """The headline steps on the fp32-MFMA kernels (CNF_STEP_FP32=1: k_step3, streamed step launches), in a child process
    started after the timed region.  Returns the child's figures, with the fp32 roofline fraction they amount to."""
    env = {k: v for k, v in os.environ.items() if not (k.startswith("ROCP") or k == "LD_PRELOAD")}
    env.update(CNF_STEP_FP32="1", HIP_FORCE_DEV_KERNARG="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.abspath(__file__), "--fp32-child", "--steps", str(max(10, min(args.steps, 50))), "--warmup", "3",
           "--batch", str(args.batch), "--depth", "1", "--no-pmc", "--no-cpu-baseline", "--no-reference-suite", "--kernel", args.kernel]
    if args.fixed_dt > 0:
        cmd += ["--fixed-dt", str(args.fixed_dt)]
    try:
        r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=180)
        line = [l for l in r.stdout.splitlines() if l.startswith("FP32CHILD ")]
        if r.returncode != 0 or not line:
            return {"error": f"child rc {r.returncode}: {(r.stderr or r.stdout)[-300:]}"}
        rec = json.loads(line[-1][len("FP32CHILD "):])
    except Exception as e:                                   # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"}
    from continuousnf.jl_amd import configs
    wl = configs.BASELINE[3]
    M = sum(a * b for a, b in zip(wl.dims[:-1], wl.dims[1:]))
    tf = rec["value"] * args.batch * (4.0 * M + 6.0 * wl.n_in) / 1e12
    rec.update({"kernel": "k_step3 (v_mfma_f32_16x16x4_f32, one step attempt per launch), CNF_STEP_FP32=1",
                "unit": "RHS-evals/s", "achieved_tflops_whole_step": tf, "frac_of_fp32_peak_whole_step": tf / PEAK_F32_TFLOPS})
    return rec


def run_rank(args):
    import torch
    import torch.distributed as dist

    import continuousnf.jl_amd as cnf
    from continuousnf.jl_amd import _lib, configs
    from continuousnf.jl_amd.parallel import RcclComm, allreduce_sums

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("CNF_BENCH_WATCHDOG"):
        # a rank that is still running after this many seconds prints every thread's Python stack and exits (rehearsals, tests)
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["CNF_BENCH_WATCHDOG"]), exit=True)
    # one rank per GPU; CNF_BENCH_BACKEND=gloo lets several ranks share one card for a rehearsal
    backend = os.environ.get("CNF_BENCH_BACKEND", "nccl")
    if os.environ.get("CNF_BENCH_DRYRUN") == "1":
        # rehearsal of the launch logic alone (tests/test_dist_gloo.py, no GPU): rendezvous, count the ranks, stop
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        n = torch.ones(1)
        if world > 1:
            dist.init_process_group("gloo", rank=rank, world_size=world)
            dist.all_reduce(n)
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"dryrun": True, "n_gpus": world, "ranks_seen": int(n), "local_rank": local_rank}), flush=True)
        return
    ndev = torch.cuda.device_count()
    if ndev < 1:
        sys.exit("bench.py: no GPU visible (the HIP backend has no CPU fallback)")
    if backend == "nccl" and world > 1 and local_rank >= ndev:
        sys.exit(f"bench.py: local rank {local_rank} has no device of its own ({ndev} visible)")
    dev_index = local_rank % ndev
    if world > ndev:
        # rehearsal with several ranks on one GPU (gloo): the one-launch solve needs every CU for itself; two processes'
        # launches would each hold CUs the other waits for -- stream step launches instead
        os.environ.setdefault("CNF_PERSISTENT", "0")
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)
    comm, collective, rccl_attempt = None, "none (1 rank)", None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        collective = f"torch.distributed all_reduce ({backend})"
        if backend == "nccl":
            # the path's own collective: an RCCL communicator owned through the C ABI (what a torch-less
            # caller uses); every rank must succeed, otherwise all fall back to torch's communicator
            ok = 1
            try:
                comm = RcclComm.from_torch_group(dev_index)
                ok = int(comm.size() == world)
            except Exception as e:                                    # noqa: BLE001
                print(f"[rank {rank}] cnf_comm_init failed: {e}", file=sys.stderr)
                ok = 0
            flag = torch.tensor([ok], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag) == 1:
                collective = "cnf_loss_allreduce (RCCL ncclAllReduce via the C ABI)"
            else:
                comm = None
        elif os.environ.get("CNF_BENCH_TRY_RCCL") == "1":
            # rehearsal on one card: the C ABI's communicator is ATTEMPTED (RCCL refuses two ranks on one GPU: "Duplicate GPU
            # detected"); whatever it answers, every rank must agree on the outcome and carry on over the rehearsal backend
            ok, why = 1, ""
            try:
                comm = RcclComm.from_torch_group(dev_index)
                ok = int(comm.size() == world)
            except Exception as e:                                    # noqa: BLE001
                ok, why = 0, f"{type(e).__name__}: {e}"
            flag = torch.tensor([ok])
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag) == 1:
                collective = "cnf_loss_allreduce (RCCL ncclAllReduce via the C ABI)"
                rccl_attempt = "accepted"
            else:
                if comm is not None:
                    comm.close()
                comm = None
                rccl_attempt = "refused, fell back to the rehearsal backend: " + why[-300:]

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

    def reduce_sums(local):
        if comm is not None:
            return comm.allreduce_sums(icnf, local)
        return allreduce_sums(local)

    def step():
        # one rank's share of `loss`: solve + post-processing + the 5 local sums in one C call, then the all-reduce
        _, _, local = cnf.inference(icnf, mode, xs, ps, {}, eps=eps, with_sums=True)
        return icnf.last_stats, reduce_sums(local)

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # hold the clocks: `--prewarm` seconds of the same steps first (the driver's K and W can be a handful: 20 solves are
    # 15 ms, which would sit on the DVFS ramp), then the W counted warm-up steps
    # (the pre-warm is bounded by each rank's own clock, so ranks run DIFFERENT numbers of it: nothing collective may be in
    # it -- a rank still all-reducing while its peer has moved on to the barrier below is a deadlock, seen in the two-rank
    # rehearsal of round 5.  Local solves only; the counted warm-up steps and the timed steps carry the all-reduce.)
    tw = time.perf_counter()
    while time.perf_counter() - tw < args.prewarm:
        cnf.inference(icnf, mode, xs, ps, {}, eps=eps, with_sums=True)
    for _ in range(args.warmup):
        step()
    sync()
    if args.fp32_child:
        # the CNF_STEP_FP32=1 leg of the parent's JSON line: the same steps on the fp32-MFMA kernels (k_step3), timed alone
        t0 = time.perf_counter(); nf = 0
        for _ in range(args.steps):
            st, _ = step(); nf += st["nf"]
        sync()
        el = time.perf_counter() - t0
        print("FP32CHILD " + json.dumps({"value": nf / el, "ms_per_step": el / args.steps * 1e3, "nf_per_solve": st["nf"],
                                         "launches_per_solve": st["launches"], "steps": args.steps}), flush=True)
        icnf.close()
        return
    # the one-launch solve adds up its own durations on the device (100 MHz real-time clock, workgroup 0, entry to exit)
    # while the timed region runs: nothing is added to the region, the total is read after it
    import ctypes as C
    _lib.check(_lib.lib().cnf_solve_kernel_time(icnf.handle(), 1, None, None), icnf.handle())
    # The K steps are SUBMITTED (cnf_inference_submit): each step's solve, post-processing, loss sums and all-reduce are
    # enqueued on the stream while the step before is still running, and collected (status, statistics) one step behind --
    # the GPU goes from one solve straight into the next instead of idling ~25 us per step while the host turns round.
    # Every step is complete, collective included, before the closing synchronisation.  --depth 1: one at a time.
    t0 = time.perf_counter()
    nf_total = 0
    if args.depth > 1:
        # (the all-reduce of a step's sums is enqueued AFTER its collect: a launch that fell back to the streamed driver
        # has its sums only then; the next step is already running behind it, so the GPU still goes from solve to solve)
        pend = []
        for _ in range(args.steps):
            _, _, local = cnf.inference_submit(icnf, mode, xs, ps, {}, eps=eps, with_sums=True)
            pend.append(local)
            if len(pend) >= args.depth:
                nf_total += cnf.inference_collect(icnf)["nf"]
                sums = reduce_sums(pend.pop(0))
        while pend:
            nf_total += cnf.inference_collect(icnf)["nf"]
            sums = reduce_sums(pend.pop(0))
        st = icnf.last_stats
    else:
        for _ in range(args.steps):
            st, sums = step()
            nf_total += st["nf"]
    sync()
    elapsed = time.perf_counter() - t0
    local_elapsed = elapsed
    k_mean_us, k_launches = C.c_float(), C.c_int()
    _lib.check(_lib.lib().cnf_solve_kernel_time(icnf.handle(), 0, C.byref(k_mean_us), C.byref(k_launches)), icnf.handle())
    # beside it, outside the timed region: the same K steps one at a time (each call returns when its solve has finished)
    one_at_a_time = None
    if args.depth > 1 and world == 1:
        t1 = time.perf_counter()
        nf1 = 0
        for _ in range(args.steps):
            st1, _ = step()
            nf1 += st1["nf"]
        sync()
        e1 = time.perf_counter() - t1
        one_at_a_time = {"value": nf1 / e1, "ms_per_step": e1 / args.steps * 1e3}
    loss = cnf.loss_from_sums(icnf, mode, sums)

    ranks_seen, per_rank_ms, allreduce_us = 1, [elapsed / args.steps * 1e3], None
    per_rank_loss = [loss]
    if world > 1:
        tdev = dev if backend == "nccl" else "cpu"
        t = torch.tensor([elapsed, float(nf_total), 1.0], dtype=torch.float64, device=tdev)
        tmax = t.clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone(); dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed, nf_all, ranks_seen = float(tmax[0]), float(tsum[1]), int(round(float(tsum[2])))
        gathered = [torch.zeros(2, dtype=torch.float64, device=tdev) for _ in range(world)]
        dist.all_gather(gathered, torch.tensor([local_elapsed / args.steps * 1e3, loss], dtype=torch.float64, device=tdev))
        per_rank_ms = [float(g[0]) for g in gathered]
        per_rank_loss = [float(g[1]) for g in gathered]             # (every rank forms the loss from the all-reduced sums)
        # the collective by itself: 5 floats, latency-bound
        probe = torch.ones(5, dtype=torch.float32, device=dev)
        for _ in range(5):
            reduce_sums(probe.clone())
        sync()
        n_ar = 50
        ta = time.perf_counter()
        for _ in range(n_ar):
            probe = reduce_sums(probe * (1.0 / world))
        torch.cuda.synchronize(dev)
        allreduce_us = (time.perf_counter() - ta) / n_ar * 1e6
    else:
        nf_all = float(nf_total)

    # ---- roofline of the dominant kernel, measured live with HIP events on the launch stream
    roof = None
    if rank == 0:
        import ctypes as C
        l, h = _lib.lib(), icnf.handle()
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
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if k_launches.value == args.steps:
                # the one-launch solve (k_solve3b) took every solve of the timed region: the dominant kernel IS the solve the
                # headline times.  Its launch = one adaptive solve of this workload; its duration = the mean the kernel
                # itself measured over the timed region (what rocprofv3 --kernel-trace reports for it); nf from the stats
                per_launch_s = k_mean_us.value * 1e-6
                units = float(st["nf"])
                kname = (f"k_solve3b: the whole Tsit5 solve in one launch ({st['nf']} RHS evaluations, "
                         f"{st['naccept'] + st['nreject']} step attempts; mean over the {k_launches.value} launches of the timed "
                         f"region, in-kernel 100 MHz clock)")
            else:
                run()
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
        # HBM bytes per launch of that kernel come from PMC counters, which need their own rocprofv3 --pmc
        # passes (FETCH_SIZE and WRITE_SIZE cannot share one; bench.py cannot profile itself): the figure is
        # REPLAYED from the committed summary of tools/collect_profiles.sh, and says so.
        traffic, traffic_source = None, None
        pmc = os.path.join(ROOT, PMC_FILE)
        one_launch = "k_solve3b" in kname
        if kernel_used == _lib.KERNEL_MFMA and B == 8192 and one_launch and world == 1 and not args.no_pmc:
            # measured in THIS run (two rocprofv3 child passes over the same solves, after the timed region)
            live, info = measure_traffic_live("k_solve3b<false")
            if live is not None:
                traffic = live
                traffic_source = (f"measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes over {info} launches of "
                                  f"the same solves (tools/prof_rhs.py bench), 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 per launch")
            else:
                traffic_source = f"live measurement unavailable ({info})"
        if traffic is None and kernel_used == _lib.KERNEL_MFMA and os.path.exists(pmc) and B == 8192:
            live_note = traffic_source
            rec = json.load(open(pmc))
            traffic = rec.get("hbm_bytes_per_launch")
            traffic_source = (f"{PMC_FILE} (rocprofv3 --pmc passes of tools/collect_profiles.sh over launches of "
                              f"'{rec.get('kernel', '?')}'; built from commit {rec.get('commit', '?')}); not measured in this run"
                              + (f" [{live_note}]" if live_note else ""))
            if ("k_solve3b" in rec.get("kernel", "")) != ("k_solve3b" in kname):
                traffic, traffic_source = None, None        # the committed counters are of the other driver's kernel
        tf = units * fl.value / per_launch_s / 1e12
        gbs = units * by.value / per_launch_s / 1e9
        # The step kernel of the headline shape forms every fp32 product from six bf16 MFMA terms on operands split EXACTLY
        # into three bf16 pieces (CNF_STEP_FP32=1 selects the fp32-MFMA kernel).  `achieved` stays what the contract
        # defines -- algorithmic fp32 flops per launch over the launch time -- and is priced against the fp32 MFMA peak;
        # `executed` prices the bf16 flops actually issued (6 per fp32 product) against the bf16 peak.
        split = kernel_used == _lib.KERNEL_MFMA and os.environ.get("CNF_STEP_FP32") != "1"
        roof = {"bound": "mfma", "achieved": tf, "peak": PEAK_F32_TFLOPS, "unit": "TFLOP/s",
                "frac": tf / PEAK_F32_TFLOPS, "traffic": traffic, "traffic_source": traffic_source, "kernel": kname,
                "arithmetic": ("fp32 result of every product = six v_mfma_f32_16x16x32_bf16 terms on operands split exactly "
                               "into three bf16 pieces, fp32 accumulate; measured as accurate as v_mfma_f32_16x16x4_f32 "
                               "(tools/ubench/bf16_split.hip, parity suite unchanged)") if split else "v_mfma_f32_16x16x4_f32",
                "executed": ({"unit": "TFLOP/s bf16", "achieved": 6.0 * tf, "peak": PEAK_BF16_TFLOPS,
                              "frac": 6.0 * tf / PEAK_BF16_TFLOPS} if split else None),
                # the same, flat: the ceiling of the instructions the kernel ISSUES, in fp32-equivalent flops (six bf16 MFMA
                # terms per fp32 product: 2500 / 6), and the fraction of it reached -- to be read beside `frac`
                "ceiling_executed": (PEAK_BF16_TFLOPS / 6.0 if split else PEAK_F32_TFLOPS),
                "frac_executed": (tf / (PEAK_BF16_TFLOPS / 6.0) if split else tf / PEAK_F32_TFLOPS),
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
            "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if os.environ.get("CNF_STEP_FP32") == "1" else "f32 (products as six bf16 MFMA terms of exactly split operands, fp32 accumulate)",
            "data": "synthetic",
            "config": {"workload": "BASELINE configs[2]: RNODE nvars=32 naugs=0, MLP 32-128-128-32 tanh, "
                                   f"batch {B} per GPU, TrainMode Hutchinson VJP, Tsit5 tspan (0,1), "
                                   + (f"fixed dt={args.fixed_dt}" if args.fixed_dt > 0 else
                                      "adaptive reltol=sqrt(eps32) abstol=eps32"),
                       "global_batch": B * world, "parallelism": f"columns sharded x{world}",
                       "kernel": {1: "generic", 2: "mfma"}.get(st["kernel_used"], "?")},
            "ranks_seen": ranks_seen, "backend": backend if world > 1 else None, "collective": collective,
            "per_rank_ms_per_step": per_rank_ms, "per_rank_loss": per_rank_loss, "allreduce_us": allreduce_us,
            "rccl_attempt": rccl_attempt,
            "sample_evals_per_s": nf_all / elapsed * B,
            "nf_per_solve": st["nf"], "naccept": st["naccept"], "nreject": st["nreject"],
            "launches_per_solve": st["launches"], "steps_in_flight": args.depth, "one_at_a_time": one_at_a_time, "loss": loss,
            "roofline": roof,
        }
        if world == 1 and roof is not None and os.environ.get("CNF_STEP_FP32") != "1" and st["kernel_used"] == _lib.KERNEL_MFMA:
            # beside the split-bf16 number, outside the timed region: the SAME steps on the exact-fp32 MFMA kernels
            # (v_mfma_f32_16x16x4_f32; CNF_STEP_FP32 is read once per process, hence a child; this process idles meanwhile)
            roof["fp32_mfma"] = fp32_child_leg(args)
        if world == 1 and not args.no_cpu_baseline:
            kwb = dict(dt=args.fixed_dt, adaptive=False) if args.fixed_dt > 0 else kw
            out["cpu_baseline"], out["cpu_other"] = cpu_baseline(B, flat, xs_h, eps_h, kwb)
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
            out["nf_gpu_vs_cpu"] = {"gpu": st["nf"], "cpu": out["cpu_baseline"].get("nf_per_solve"),
                                    "note": "same workload and tolerances; the error estimate is at round-off level here, so the "
                                            "step count differs by an attempt between arithmetics (profiles/round3_step_trace.md)"}
        if world == 1 and not args.no_reference_suite:
            out["reference_suite"] = reference_suite()
            out["gradient"] = gradient_leg()
        print(json.dumps(out), flush=True)
    if comm is not None:
        comm.close()
    icnf.close()
    if world > 1:
        dist.destroy_process_group()


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args)
    run_rank(args)


if __name__ == "__main__":
    main()
