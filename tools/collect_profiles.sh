#!/bin/bash
# Runs on the GPU box (inside gpurun): rocprofv3 summaries for profiles/.
#  1. kernel trace + stats of the default bench command
#  2. HBM traffic counters of the fused step kernel (separate --pmc passes, no trace domains
#     beyond --kernel-trace), on tools/prof_rhs.py (fixed-dt solve = step launches only)
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profiles_r1
mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -- python3 bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1
echo "bench rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 tools/prof_rhs.py step 16 > $OUT/pmc_fetch.log 2>&1
echo "fetch rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 tools/prof_rhs.py step 16 > $OUT/pmc_write.log 2>&1
echo "write rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_sq -- python3 tools/prof_rhs.py step 16 > $OUT/pmc_sq.log 2>&1
echo "sq rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/grad -- python3 tools/prof_grad.py 3 8192 3 > $OUT/grad.log 2>&1
echo "grad rc=$?"
