#!/bin/bash
# Runs on the GPU box: where do the LDS bank conflicts of the headline kernel come from?  
# k_solve3b (fixed dt, 32 steps = 193 evaluations per launch) in three builds -- as shipped, without the epilogue's split-image
# stores (-DS3_ABL_NOSTORE) and without the operand reads (-DS3_ABL_NOLOAD) -- under rocprofv3 --pmc (SQ LDS counters), plus a
# --kernel-trace --stats pass each for the duration.  The ablated builds compute garbage; the counters and the time are what is read.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export HIP_FORCE_DEV_KERNARG=1
OUT=gpurun_out/lds_abl
mkdir -p $OUT
for v in base NOSTORE NOLOAD; do
  if [ $v = base ]; then unset CNFHIP_LIB; else export CNFHIP_LIB=$PWD/build_abl/lib_$v.so; fi
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$v/stats -- python3 tools/prof_rhs.py step 32 > $OUT/$v.log 2>&1
  timeout -k 10 120 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS --output-format csv -d $OUT/$v/sq -- python3 tools/prof_rhs.py step 32 > $OUT/$v.sq.log 2>&1
  echo "$v done: $(grep '^step' $OUT/$v.log)"
done
