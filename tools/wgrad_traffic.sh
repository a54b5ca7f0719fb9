#!/bin/bash
# HBM bytes fetched per launch of the weight-gradient contraction (rocprofv3 --pmc FETCH_SIZE, its own pass) on the gradient
# path of config 3: tells re-reads that hit an L2 from re-reads that cross the fabric.  FETCH_SIZE is in units of 64 bytes...
# see /opt/skills/guides/MI355X_MICROARCH.md for the unit; printed raw and as bytes at 32 B and 64 B per unit.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
rm -rf /tmp/wgt
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/wgt -o r -- python3 $R/tools/prof_grad.py 3 8192 2 > /tmp/wgt.log 2>&1 || { tail -5 /tmp/wgt.log; exit 1; }
f=$(find /tmp/wgt -name '*counter_collection.csv' | head -1)
python3 - "$f" <<'P'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "wgrad" in n or "k_adj3" in n:
        acc[n[:30]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, d in acc.items():
    for c, v in d.items():
        print("%-32s %-14s launches %3d  mean %.4g  max %.4g" % (n, c, len(v), sum(v) / len(v), max(v)))
P
