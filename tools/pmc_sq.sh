#!/bin/bash
# usage: tools/pmc_sq.sh <workload> <tag>  — instruction-mix / wave-stall counter passes (SQ block) for the dominant kernel.
# At most four SQ counters per pass, every pass under its own timeout (see pmc3.sh).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
W=$1; TAG=$2
mkdir -p gpurun_out/$TAG
run() { timeout -k 10 150 rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/$TAG/$1 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-api --no-variants $BENCH_EXTRA --workload $W > gpurun_out/$TAG/$1.log 2>&1 || { echo "pass $1 failed"; return 1; }; echo "pass $1 ok"; }
run SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU || exit 1
run SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM || exit 1
run SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY || exit 1
run SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_INST_CYCLES_SALU || exit 1
run SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_BRANCH || exit 1
python3 tools/pmc_summary.py gpurun_out/$TAG > /dev/null
