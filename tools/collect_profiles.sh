#!/bin/bash
# Runs on the GPU box (inside gpurun): rocprofv3 summaries for profiles/.
#  1. kernel trace + stats of the default bench command
#  2. counters of the headline kernel in separate --pmc passes (each with --kernel-trace only: FETCH_SIZE and WRITE_SIZE
#     cannot share a pass), on tools/prof_rhs.py bench (the adaptive solves bench.py times: with the one-launch solve
#     k_solve3b a launch IS such a solve; CNF_PERSISTENT=0: the step launches of the streamed solve)
#  3. kernel split of the gradient path
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export HIP_FORCE_DEV_KERNARG=1
TAG=${1:-r3}
OUT=gpurun_out/profiles_$TAG
mkdir -p $OUT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/bench -- python3 bench.py --no-cpu-baseline --no-pmc > $OUT/bench_under_rocprof.log 2>&1
echo "bench rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 tools/prof_rhs.py bench 16 > $OUT/pmc_fetch.log 2>&1
echo "fetch rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 tools/prof_rhs.py bench 16 > $OUT/pmc_write.log 2>&1
echo "write rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES --output-format csv -d $OUT/pmc_sq -- python3 tools/prof_rhs.py bench 16 > $OUT/pmc_sq.log 2>&1
echo "sq rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_wait -- python3 tools/prof_rhs.py bench 16 > $OUT/pmc_wait.log 2>&1
echo "wait rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/grad -- python3 tools/prof_grad.py 3 8192 3 > $OUT/grad.log 2>&1
echo "grad rc=$?"
# the same network at the reference's training batch (two-launch pullback), and config 5's network (generic MFMA pullback)
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/grad_b32 -- python3 tools/prof_grad.py 3 32 6 > $OUT/grad_b32.log 2>&1
echo "grad_b32 rc=$?"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/grad_cfg5 -- python3 tools/prof_grad.py 5 2048 4 > $OUT/grad_cfg5.log 2>&1
echo "grad_cfg5 rc=$?"
timeout -k 10 200 python3 tools/prof_grad.py > $OUT/grad_table.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/testmode -- python3 tools/prof_testmode.py > $OUT/testmode.log 2>&1
echo "testmode rc=$?"
# every BASELINE configuration at full size (uses the oracle as the checker: tests/measure_configs.py)
timeout -k 10 500 python3 tests/measure_configs.py > $OUT/all_configs.log 2>&1
echo "all_configs rc=$?"
grep "^{" $OUT/all_configs.log > $OUT/all_configs.jsonl
timeout -k 10 200 python3 tools/prof_testmode.py > $OUT/testmode_plain.log 2>&1
