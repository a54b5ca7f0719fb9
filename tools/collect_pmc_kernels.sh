#!/bin/bash
# Runs on the GPU box (inside gpurun): rocprofv3 passes for the kernels BESIDE the headline one:
# k_solve_wave (config 2), k_solve_bcast (config 5, Train and exact trace), k_trace3s<SOLVE> (TestMode of the headline network:
# tools/prof_testmode.py), k_adj3 + k_wgrad_wave (the gradient at config 3).  Per program: one --kernel-trace --stats pass and
# three --pmc passes (FETCH_SIZE and WRITE_SIZE each alone, then the SQ set), every pass with --kernel-trace only beside it.
# The program goes directly after `--`.  tools/summarize_pmc_kernels.py turns the result into profiles/<name>_kernel_rooflines.json.
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export HIP_FORCE_DEV_KERNARG=1
TAG=${1:-r5}
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
SQ="SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
run_set() {       # name, then the program and its arguments
    local name=$1; shift
    timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$name/stats -- "$@" > $OUT/$name.log 2>&1 || return 1
    echo "$name stats rc=$?"
    timeout -k 10 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/$name/fetch -- "$@" > $OUT/$name.fetch.log 2>&1 || return 1
    timeout -k 10 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/$name/write -- "$@" > $OUT/$name.write.log 2>&1 || return 1
    timeout -k 10 150 rocprofv3 --kernel-trace --pmc $SQ --output-format csv -d $OUT/$name/sq -- "$@" > $OUT/$name.sq.log 2>&1 || return 1
    echo "$name pmc done"
}
if [ "${2:-all}" = grad ]; then        # only the gradient programs (the other directories of the tag stay as they are)
run_set cfg3_grad python3 tools/prof_grad.py 3 8192 4 &&
run_set cfg3_b32_grad python3 tools/prof_grad.py 3 32 8 &&
run_set cfg5_grad python3 tools/prof_grad.py 5 2048 4
echo "grad rc=$?"
exit 0
fi
run_set cfg2_train python3 tools/prof_solve.py 2 train 16 &&
run_set cfg1_train python3 tools/prof_solve.py 1 train 16 &&
run_set cfg5_train python3 tools/prof_solve.py 5 train 16 &&
run_set cfg5_test python3 tools/prof_solve.py 5 test 16 &&
run_set cfg5_jvp python3 tools/prof_solve.py 5 jvp 16 &&
run_set cfg3_test python3 tools/prof_solve.py 3 test 6 &&
run_set cfg3_grad python3 tools/prof_grad.py 3 8192 4 &&
run_set cfg3_b32_grad python3 tools/prof_grad.py 3 32 8 &&
run_set cfg5_grad python3 tools/prof_grad.py 5 2048 4
echo "all rc=$?"
