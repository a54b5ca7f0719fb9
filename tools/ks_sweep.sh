cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for ks in 64 80 96 112 128; do
  export CNF_WGRAD_KS=$ks
  rm -rf /tmp/ks_$ks
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$ks -o r -- python3 $R/tools/prof_grad.py 3 8192 4 > /tmp/ks_$ks.log 2>&1
  f=$(find /tmp/ks_$ks -name '*kernel_stats*' | head -1)
  echo "ks $ks: $(grep cfg3 /tmp/ks_$ks.log)"; grep wgrad $f | cut -d, -f2-4,6,7
done
