"""gpurun_out/pmc_<tag> (tools/collect_pmc_kernels.sh) -> profiles/<name>_kernel_rooflines.json + .md: one roofline line per kernel
BESIDE the headline one.  Per program the dominant kernel's mean duration comes from the --kernel-trace --stats pass; FETCH_SIZE,
WRITE_SIZE and the SQ set each from their own --pmc pass (means over the launches of that kernel); algorithmic flops / bytes from
the program's own JSON line (cnf_rhs_work x evaluations per solve) or, for the gradient, from the model of DESIGN 4.4.
HBM traffic = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024 (the guide's gfx950 read-side correction for 16-byte loads; KiB units).
    python tools/summarize_pmc_kernels.py [tag=r5] [name=round5]"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r5"
name = sys.argv[2] if len(sys.argv) > 2 else "round5"
SRC = os.path.join(ROOT, "gpurun_out", f"pmc_{tag}")
PEAK_F32, PEAK_HBM = 157.3e12, 8.0e12
N_SIMD, CLK = 1024, 2.4e9


def short(k):
    k = k.replace("(anonymous namespace)::", "").replace("void ", "")
    return k.split("(")[0]


def stats(prog):
    f = glob.glob(os.path.join(SRC, prog, "stats", "*", "*kernel_stats.csv"))
    rows = list(csv.DictReader(open(f[0]))) if f else []
    return [(short(r["Name"]), int(r["Calls"]), float(r["AverageNs"]) * 1e-3, float(r["Percentage"])) for r in rows]


def counters(prog, sub, kname):
    f = glob.glob(os.path.join(SRC, prog, sub, "*", "*counter_collection.csv"))
    if not f:
        return {}
    acc, n = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(f[0])):
        if short(r["Kernel_Name"]) == kname:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    return {c: acc[c] / n[c] for c in acc}


def line_of(prog):
    p = os.path.join(SRC, prog + ".log")
    for l in open(p):
        if l.startswith("{"):
            return json.loads(l)
    return None


out = []
for prog in sorted(os.listdir(SRC)):
    if not os.path.isdir(os.path.join(SRC, prog)):
        continue
    ks = stats(prog)
    if not ks:
        continue
    info = line_of(prog)
    # the kernels of weight: everything above 8 % of the program's GPU time (the gradient has two)
    for kname, calls, avg_us, pct in ks:
        if pct < 8.0 and not (prog.endswith("_grad") and kname.startswith("k_adj")):      # (the pullback's parallel launch is short)
            continue
        c = {}
        for sub in ("fetch", "write", "sq"):
            c.update(counters(prog, sub, kname))
        rec = {"program": prog, "kernel": kname, "launches_in_stats_pass": calls, "mean_us": round(avg_us, 2), "share_of_gpu_time_pct": pct,
               "counters_per_launch": {k: round(v, 1) for k, v in sorted(c.items())}}
        if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
            rec["hbm_bytes_per_launch"] = 2 * c["FETCH_SIZE"] * 1024 + c["WRITE_SIZE"] * 1024
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            # the counter adds up busy cycles over the SIMDs (x4 per CU: the round-4 reading, SQ_INSTS_MFMA-consistent)
            rec["mfma_pipe_busy_frac"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (N_SIMD * avg_us * 1e-6 * CLK), 3)
        if c.get("SQ_LDS_IDX_ACTIVE"):
            rec["lds_bank_conflict_share"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 3)
        if c.get("SQ_INSTS_MFMA"):
            rec["valu_per_mfma"] = round(c.get("SQ_INSTS_VALU", 0.0) / c["SQ_INSTS_MFMA"], 2)
        if info and "k_solve" in kname or (info and "k_trace3s" in kname and calls <= info["solves"] + 2):
            fl, by = info["nf"] * info["flops_per_rhs"], info["nf"] * info["bytes_per_rhs"]
            rec.update({"workload": f"config {info['cfg']} {info['mode']}, B = {info['B']}, adaptive, nf = {info['nf']} per launch",
                        "algorithmic_flops_per_launch": fl, "algorithmic_bytes_per_launch": by,
                        "achieved_TFLOPs": round(fl / (avg_us * 1e-6) / 1e12, 2), "frac_of_fp32_peak": round(fl / (avg_us * 1e-6) / PEAK_F32, 3),
                        "achieved_GBs_algorithmic": round(by / (avg_us * 1e-6) / 1e9, 1), "frac_of_hbm_peak": round(by / (avg_us * 1e-6) / PEAK_HBM, 4),
                        "us_per_evaluation": round(avg_us / info["nf"], 2)})
            if info["mode"] == "test" and len(info["dims"]) > 3:
                d = info["dims"]
                M = sum(a * b for a, b in zip(d[:-1], d[1:]))
                ex = info["B"] * (2.0 * M + 2.0 * d[0] * d[1] * d[2] + 2.0 * d[2] * d[3]) * info["nf"]
                rec["executed_formulation"] = {"note": "re-associated trace tr(D3 W3 D2 (W2 D1 W1)): forward + one d2 x d1 x d0 product per sample + the "
                                                       "contraction; cnf_rhs_work prices the reference's n_in tangent sweeps, which the kernel does not execute",
                                               "flops_per_launch": ex, "achieved_TFLOPs": round(ex / (avg_us * 1e-6) / 1e12, 2),
                                               "frac_of_split_bf16_ceiling_416.7": round(ex / (avg_us * 1e-6) / (2500e12 / 6), 3)}
        elif prog.endswith("_grad") and kname.startswith(("k_adj", "k_wgrad")):
            # the gradient programs (tools/prof_grad.py cfg B): "cfgN B=...: ... steps S+R" in the log; the pullback model of DESIGN 4.4:
            # a stage pullback is four sweeps of 2M flops per sample (three in the stage-parallel launch, one in the sequential one)
            import re
            txt = open(os.path.join(SRC, prog + ".log")).read()
            mm = re.search(r"cfg(\d) B=(\d+):.*steps (\d+)\+", txt)
            cfgn, B, steps = int(mm.group(1)), int(mm.group(2)), int(mm.group(3))
            dims = {3: (32, 128, 128, 32), 5: (128, 384, 128)}[cfgn]
            M = sum(x * y for x, y in zip(dims[:-1], dims[1:]))
            P = M + sum(dims[1:])
            # launches of this kernel per gradient = calls / gradients; prof_grad runs loss_and_grad (reps + 1) times
            grads = {"cfg3_grad": 5, "cfg3_b32_grad": 9, "cfg5_grad": 5}.get(prog, 5)
            lpg = calls / grads
            stage_samples = steps * 6 * B / lpg               # (stage, sample) pairs one launch covers
            if kname.startswith("k_wgrad"):
                fl = stage_samples * 4.0 * P
                rec["workload"] = f"config {cfgn}'s network gradient, B = {B}: weight-gradient contraction, {lpg:.1f} launches per gradient"
            else:
                sweeps = 3 if ("<1>" in kname or ", 1>" in kname) else (1 if ("<2>" in kname or ", 2>" in kname) else 4)
                fl = stage_samples * sweeps * 2.0 * M
                form = {4: "one launch: all four sweeps", 3: "stage-parallel launch: sweeps 1-3", 1: "sequential launch: the hbar chain"}[sweeps]
                rec["workload"] = f"config {cfgn}'s network gradient, B = {B}, {steps} steps: {form}, {lpg:.1f} launches per gradient"
            rec.update({"algorithmic_flops_per_launch": fl, "achieved_TFLOPs": round(fl / (avg_us * 1e-6) / 1e12, 2),
                        "frac_of_fp32_peak": round(fl / (avg_us * 1e-6) / PEAK_F32, 3)})
            if "hbm_bytes_per_launch" in rec:
                rec["hbm_GBs"] = round(rec["hbm_bytes_per_launch"] / (avg_us * 1e-6) / 1e9, 1)
                rec["frac_of_hbm_peak_measured_traffic"] = round(rec["hbm_bytes_per_launch"] / (avg_us * 1e-6) / PEAK_HBM, 3)
        out.append(rec)

dst = os.path.join(ROOT, "profiles", f"{name}_kernel_rooflines.json")
json.dump({"source": f"tools/collect_pmc_kernels.sh {tag} on one MI355X; rocprofv3 --kernel-trace --stats and three --pmc passes per program",
           "peaks": {"fp32_mfma_TFLOPs": 157.3, "hbm_TBs": 8.0}, "kernels": out}, open(dst, "w"), indent=1)
md = [f"# {name}: one roofline line per kernel beside the headline one (`tools/collect_pmc_kernels.sh`, `tools/summarize_pmc_kernels.py`)", "",
      "| kernel | workload | mean µs | µs / eval | TFLOP/s (alg.) | of fp32 peak | MFMA pipe busy | VALU / MFMA | LDS conflict share | HBM MB / launch (PMC) | alg. MB |",
      "|---|---|---|---|---|---|---|---|---|---|---|"]
for r in out:
    if "executed_formulation" in r:                       # the reference formulation's flop count is not what the kernel executes
        ex = r["executed_formulation"]
        r = dict(r, achieved_TFLOPs=f"{ex['achieved_TFLOPs']} (executed formulation; {r['achieved_TFLOPs']} priced as the reference's sweeps)",
                 frac_of_fp32_peak=f"{ex['frac_of_split_bf16_ceiling_416.7']} of the split-bf16 ceiling 416.7")
    md.append("| `{}` | {} | {} | {} | {} | {} | {} | {} | {} | {} | {} |".format(
        r["kernel"][:60], r.get("workload", r["program"]), r["mean_us"], r.get("us_per_evaluation", ""), r.get("achieved_TFLOPs", ""),
        r.get("frac_of_fp32_peak", ""), r.get("mfma_pipe_busy_frac", ""), r.get("valu_per_mfma", ""), r.get("lds_bank_conflict_share", ""),
        round(r["hbm_bytes_per_launch"] / 1e6, 2) if "hbm_bytes_per_launch" in r else "",
        round(r["algorithmic_bytes_per_launch"] / 1e6, 2) if "algorithmic_bytes_per_launch" in r else ""))
open(os.path.join(ROOT, "profiles", f"{name}_kernel_rooflines.md"), "w").write("\n".join(md) + "\n")
print("\n".join(md))
