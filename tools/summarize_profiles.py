"""Turns gpurun_out/profiles_<tag> (written by tools/collect_profiles.sh on the GPU box) into the committed summaries
under profiles/.     python tools/summarize_profiles.py [tag=r2] [name=round2]

Launches of the step kernel that exit at once (queued past the end of a solve, or behind a controller that has just
finished it) are dropped before any counter is averaged: they are identified by their duration in the kernel trace
of the same pass (joined on the dispatch id), so the per-launch figures are those of FULL steps."""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r3"
name = sys.argv[2] if len(sys.argv) > 2 else "round3"
SRC = os.path.join(ROOT, "gpurun_out", f"profiles_{tag}")
DST = os.path.join(ROOT, "profiles")
os.makedirs(DST, exist_ok=True)
STEP = ("k_step3", "k_mfma")          # the step kernel of the headline shape (second / first generation)
FULL_US = 30.0                        # anything shorter did not do a full step: early exits take 6-12 us, the
                                      # single-evaluation launches of the automatic initial dt 19 us, a step 50 us


def is_solve(kname):
    # the one-launch solve (the plain instantiation, not the recording one of the gradient path): when it ran, IT is the
    # headline kernel (a launch = a whole solve)
    return "k_solve3b<false" in kname or kname.startswith("k_solve3b(")      # (<false>: round 2; <false, false>: RECORD, MULTI)


def is_step(kname):
    if HAVE_SOLVE:
        return is_solve(kname)
    return is_step_launch(kname)


def is_step_launch(kname):
    # k_step3(...) or the STEP = true instantiation k_mfma<Layout, true>(...) -- not the plain RHS kernel k_mfma<Layout, false>
    # (k_step3j is the JVP / FFJORD step kernel: not the kernel the headline bench runs)
    return kname.startswith(("k_step3(", "k_step3b(")) or "k_step3<" in kname or ("k_mfma<" in kname and ", true>(" in kname)


def first(pattern):
    g = glob.glob(os.path.join(SRC, pattern), recursive=True)
    return max(g, key=os.path.getmtime) if g else None      # newest run


def commit():
    try:
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    except OSError:
        return "?"


HAVE_SOLVE = False
_tr0 = None
for _g in glob.glob(os.path.join(SRC, "bench/**/*kernel_trace.csv"), recursive=True):
    _tr0 = _g if _tr0 is None or os.path.getmtime(_g) > os.path.getmtime(_tr0) else _tr0
if _tr0:
    HAVE_SOLVE = any(is_solve(r["Kernel_Name"]) for r in csv.DictReader(open(_tr0)))
st = first("bench/**/*kernel_stats.csv")
if st:
    rows = list(csv.DictReader(open(st)))
    with open(os.path.join(DST, f"{name}_bench_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows[:12]:
            w.writerow([r["Name"][:110], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                        r["MinNs"], r["MaxNs"]])
    log = os.path.join(SRC, "bench_under_rocprof.log")
    for line in open(log):
        if line.startswith("{"):
            open(os.path.join(DST, f"{name}_bench_under_rocprof.json"), "w").write(line)
    # duration split of the step kernel (full steps vs early exits)
    tr = first("bench/**/*kernel_trace.csv")
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(tr)) if is_step(r["Kernel_Name"])]
    full = [x for x in d if x > FULL_US]
    with open(os.path.join(DST, f"{name}_step_kernel_durations.txt"), "w") as f:
        if HAVE_SOLVE:
            f.write(f"k_solve3b launches (one adaptive solve of the bench workload each): {len(d)}\n")
            if d:
                f.write(f"per launch: mean {sum(d) / len(d):.2f} us, min {min(d):.2f}, max {max(d):.2f}\n")
        else:
            f.write(f"fused step kernel launches: {len(d)}; doing a full step: {len(full)}; "
                    f"others (early exits of launches queued past the end, single-evaluation launches of the automatic "
                    f"initial dt): {len(d) - len(full)}\n")
            if full:
                f.write(f"full-step launches: mean {sum(full) / len(full):.2f} us, min {min(full):.2f}, max {max(full):.2f}\n")


def pmc(dirname):
    """mean counter values over the FULL-step launches of this pass"""
    f = first(f"{dirname}/**/*counter_collection.csv")
    t = first(f"{dirname}/**/*kernel_trace.csv")
    if not f:
        return {}, 0, 0
    dur = {}
    if t:
        for r in csv.DictReader(open(t)):
            dur[r.get("Dispatch_Id")] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg, seen, kept = collections.defaultdict(list), set(), set()
    for r in csv.DictReader(open(f)):
        if not is_step(r["Kernel_Name"]):
            continue
        did = r.get("Dispatch_Id")
        seen.add(did)
        if did in dur and dur[did] < FULL_US:
            continue
        kept.add(did)
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}, len(seen), len(kept)


out = {"kernel": ("k_solve3b (one adaptive Tsit5 solve of the bench workload per launch, six-term bf16 products, B = 8192)" if HAVE_SOLVE else
                  "k_step3b (fused Tsit5 step of the headline shape on six-term bf16 products, B = 8192)"),
       "commit": commit(), "per_launch": {}, "launches": {}}
for d in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_wait"):
    vals, n, k = pmc(d)
    out["per_launch"].update(vals)
    out["launches"][d] = {"step_launches": n, "full_steps_averaged": k}
pl = out["per_launch"]
if "FETCH_SIZE" in pl and "WRITE_SIZE" in pl:
    # MI355X_MICROARCH.md HBM section: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half the bytes
    # of wide (16 B/lane) coalesced reads -> doubled; WRITE_SIZE exact.
    out["hbm_bytes_per_launch"] = 2 * pl["FETCH_SIZE"] * 1024 + pl["WRITE_SIZE"] * 1024
    out["note"] = ("traffic = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950 read-side correction); the state loads are "
                   "4 B/lane, a width the guide calls uncalibrated, so the read side is an upper bound; early-exit "
                   "launches (< 15 us) are excluded from every mean")
json.dump(out, open(os.path.join(DST, f"{name}_step_kernel_pmc.json"), "w"), indent=1)
print(json.dumps(out, indent=1))

# gradient path (row f3): kernel split of loss + loss_and_grad at config 3, B = 8192
gs = first("grad/**/*kernel_stats.csv")
if gs:
    rows = list(csv.DictReader(open(gs)))
    with open(os.path.join(DST, f"{name}_grad_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows[:12]:
            w.writerow([r["Name"][:110], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                        r["MinNs"], r["MaxNs"]])

# TestMode of the headline network (exact trace): kernel split of tools/prof_testmode.py
ts = first("testmode/**/*kernel_stats.csv")
if ts:
    rows = list(csv.DictReader(open(ts)))
    with open(os.path.join(DST, f"{name}_testmode_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows[:8]:
            w.writerow([r["Name"][:110], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                        r["MinNs"], r["MaxNs"]])
for src, dst in (("all_configs.jsonl", f"{name}_all_configs.jsonl"), ("testmode_plain.log", f"{name}_testmode.txt")):
    pth = os.path.join(SRC, src)
    if os.path.exists(pth):
        open(os.path.join(DST, dst), "w").write("".join(l for l in open(pth) if l.startswith(("{", "cfg"))))
