"""Per-kernel means of the counters of one rocprofv3 --pmc pass (csv output directory).   python tools/pmc_summary.py DIR [substring]"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:70]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k in acc:
    if sub in k:
        print(k, {c: round(v / n[k][c]) for c, v in acc[k].items()}, "launches", max(n[k].values()))
