#!/bin/bash
# times the gradient path's kernels (rocprofv3 kernel stats of tools/prof_grad.py 3 8192) for several builds of the library on ONE box
# usage: tools/ab_adj.sh build_abl/lib_a.so build_abl/lib_b.so ...   (the default build is always run first)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for lib in default "$@"; do
  name=$(basename $lib .so)
  if [ "$lib" = default ]; then unset CNFHIP_LIB; else export CNFHIP_LIB=$R/$lib; fi
  rm -rf /tmp/abadj_$name
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abadj_$name -o r -- python3 $R/tools/prof_grad.py 3 8192 4 > /tmp/abadj_$name.log 2>&1 || { tail -5 /tmp/abadj_$name.log; exit 1; }
  grep cfg3 /tmp/abadj_$name.log
  f=$(find /tmp/abadj_$name -name '*kernel_stats*' | head -1); [ -z "$f" ] && find /tmp/abadj_$name | head
  echo "== $name"; python3 - "$f" <<'P'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r["Name"] for k in ("k_adj3", "k_wgrad", "k_solve3b<true")):
        print("   %-40s calls %4s avg %9.1f us  min %9.1f  max %9.1f" % (r["Name"][:40], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
P
done
