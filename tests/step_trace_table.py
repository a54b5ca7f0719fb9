#!/usr/bin/env python3
"""Per-attempt (t, h, EEst, accepted) of the bench workload (bench.py's inputs: BASELINE config 3, B = 8192, README
tolerances) from the GPU's one-launch solve (cnf_set_step_trace) and from the float32 C oracle (oracle/cnf_oracle.c), side
by side: where the two step sequences part and by how much.  Writes a markdown table.
Lives under tests/ because it uses the oracle as its checker.   usage: python tests/step_trace_table.py [out.md] [B]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")


def main():
    import torch

    import continuousnf.jl_amd as cnf
    from continuousnf.jl_amd import configs
    from oracle import c_oracle as CO
    from oracle import cnf_oracle as O

    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "step_trace.md")
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
    wl = configs.BASELINE[3]
    flat = configs.glorot_params(wl.dims, 12345)
    xs, eps = configs.synthetic_inputs(wl, B, 1)
    kw = dict(configs.README_TOLERANCES)
    rows = {}
    for jvp in (False, True):
        ic = configs.build(wl, kernel="mfma", jvp=jvp, sol_kwargs=kw)
        tr = ic.set_step_trace(64)
        dev = lambda a: torch.from_numpy(np.ascontiguousarray(a.T)).cuda().t()
        cnf.inference(ic, cnf.TrainMode(), dev(xs), flat, {}, eps=dev(eps))
        st = dict(ic.last_stats)
        n = st["naccept"] + st["nreject"]
        rows["gpu_jvp" if jvp else "gpu_vjp"] = (tr.cpu().numpy()[:n].copy(), st)
        ic.close()
    cfg, _, _ = O.baseline_cfg(3)
    ctr = np.zeros((64, 4), np.float32)
    _, cst = CO.solve(cfg, flat, O.inference_u0(cfg, xs, True), eps, True, trace=ctr, **kw)
    c = ctr[: cst["naccept"] + cst["nreject"]]
    g, gst = rows["gpu_vjp"]
    gj, gjst = rows["gpu_jvp"]
    with open(out, "w") as f:
        f.write(f"# Step sequences of the bench workload (config 3, B = {B}, reltol = sqrt(eps32), abstol = eps32)\n\n")
        f.write(f"GPU one-launch solve, VJP handle (`k_solve3b`, split-bf16 products): nf = {gst['nf']}, {gst['naccept']} accepted, "
                f"{gst['nreject']} rejected.  JVP handle (`k_solve3jb`): nf = {gjst['nf']}, {gjst['naccept']} / {gjst['nreject']}.  "
                f"C oracle (`oracle/cnf_oracle.c`, fp32 scalar loops, VJP): nf = {cst['nf']}, {cst['naccept']} accepted, "
                f"{cst['nreject']} rejected.\n\n")
        f.write("| attempt | GPU t | GPU h | GPU EEst | acc | C t | C h | C EEst | acc | EEst GPU/C | JVP-kernel EEst |\n|---|---|---|---|---|---|---|---|---|---|---|\n")
        for i in range(max(len(g), len(c), len(gj))):
            a = g[i] if i < len(g) else None
            b = c[i] if i < len(c) else None
            j = gj[i] if i < len(gj) else None
            fa = f"{a[0]:.6f} | {a[1]:.6e} | {a[2]:.4e} | {int(a[3])}" if a is not None else " | | | "
            fb = f"{b[0]:.6f} | {b[1]:.6e} | {b[2]:.4e} | {int(b[3])}" if b is not None else " | | | "
            ratio = f"{a[2] / b[2]:.4f}" if a is not None and b is not None and b[2] > 0 and abs(a[0] - b[0]) < 1e-6 * max(1, abs(b[0])) else "(different t)"
            fj = f"{j[2]:.4e}" if j is not None else ""
            f.write(f"| {i} | {fa} | {fb} | {ratio} | {fj} |\n")
    print(open(out).read())


if __name__ == "__main__":
    main()
