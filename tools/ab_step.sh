#!/bin/bash
# A/B of step-kernel builds ON ONE BOX (clocks differ by several percent between boxes and runs): every library given on
# the command line is timed in turn, several rounds, on the fixed-dt step benchmark of tools/prof_rhs.py.
#   tools/ab_step.sh build_abl/libA.so build_abl/libB.so ...
set -u
export HIP_FORCE_DEV_KERNARG=1
ROUNDS=${ROUNDS:-4}
for r in $(seq 1 $ROUNDS); do
  for lib in "$@"; do
    t=$(CNFHIP_LIB=$PWD/$lib python tools/prof_rhs.py step 512 2>/dev/null | grep "^step:" | sed 's/step: \([0-9.]*\) us.*/\1/')
    echo "round $r $(basename $lib) $t"
  done
done | tee gpurun_out/ab_step.log
python3 - <<'PY'
import collections, statistics
d = collections.defaultdict(list)
for line in open("gpurun_out/ab_step.log"):
    p = line.split()
    if len(p) == 4:
        d[p[2]].append(float(p[3]))
for k, v in d.items():
    print(f"{k}: min {min(v):.2f} median {statistics.median(v):.2f} us per step ({len(v)} runs)")
PY
