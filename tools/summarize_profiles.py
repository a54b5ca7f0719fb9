"""Turns gpurun_out/profiles_r1 (written by tools/collect_profiles.sh on the GPU box) into the
committed summaries under profiles/."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "profiles_r1")
DST = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "round1"
os.makedirs(DST, exist_ok=True)


def first(pattern):
    g = glob.glob(os.path.join(SRC, pattern), recursive=True)
    return max(g, key=os.path.getmtime) if g else None      # newest run


st = first("bench/**/*kernel_stats.csv")
if st:
    rows = list(csv.DictReader(open(st)))
    with open(os.path.join(DST, f"{tag}_bench_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows[:12]:
            w.writerow([r["Name"][:110], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                        r["MinNs"], r["MaxNs"]])
    log = os.path.join(SRC, "bench_under_rocprof.log")
    for line in open(log):
        if line.startswith("{"):
            open(os.path.join(DST, f"{tag}_bench_under_rocprof.json"), "w").write(line)
    # duration histogram of the step kernel (full steps vs early exits)
    tr = first("bench/**/*kernel_trace.csv")
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(tr))
         if "k_mfma" in r["Kernel_Name"] and "true" in r["Kernel_Name"]]
    full = [x for x in d if x > 20]
    with open(os.path.join(DST, f"{tag}_step_kernel_durations.txt"), "w") as f:
        f.write(f"fused step kernel launches: {len(d)}; doing a full step: {len(full)}; "
                f"early exits (solve already finished): {len(d) - len(full)}\n")
        if full:
            f.write(f"full-step launches: mean {sum(full) / len(full):.2f} us, min {min(full):.2f}, max {max(full):.2f}\n")


def pmc(dirname, want):
    f = first(f"{dirname}/**/*counter_collection.csv")
    if not f:
        return {}
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "k_mfma" in r["Kernel_Name"] and "true" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items() if k in want or not want}


fetch = pmc("pmc_fetch", {"FETCH_SIZE"})
write = pmc("pmc_write", {"WRITE_SIZE"})
sq = pmc("pmc_sq", set())
out = {"kernel": "k_mfma<StLayout<1,32,128,128,32>, true> (fused Tsit5 step, B=8192)",
       "per_launch": {**fetch, **write, **sq}}
if fetch and write:
    # MI355X_MICROARCH.md HBM section: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
    # reports half the bytes of wide (16 B/lane) coalesced reads -> doubled; WRITE_SIZE exact.
    out["hbm_bytes_per_launch"] = 2 * fetch["FETCH_SIZE"] * 1024 + write["WRITE_SIZE"] * 1024
    out["note"] = ("traffic = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950 read-side correction); the state "
                   "loads are 4 B/lane, a width the guide calls uncalibrated, so the read side is an upper bound")
json.dump(out, open(os.path.join(DST, f"{tag}_step_kernel_pmc.json"), "w"), indent=1)
print(json.dumps(out, indent=1))

# gradient path (row f3): kernel split of loss + loss_and_grad at config 3, B = 8192
gs = first("grad/**/*kernel_stats.csv")
if gs:
    rows = list(csv.DictReader(open(gs)))
    with open(os.path.join(DST, f"{tag}_grad_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in rows[:12]:
            w.writerow([r["Name"][:110], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"],
                        r["MinNs"], r["MaxNs"]])
