#!/bin/bash
# usage: tools/pmc3.sh <workload> <tag>  — stall-oriented counter passes (TCP / TA / TD / TCC / LDS) for the dominant kernel.
# At most four counters of one TCP / TA / TD / TCC block per pass (more aborts rocprofv3 on gfx950; the SQ block took the six
# of the last pass, profiles/r02/config3_pmc_stalls.json has all of them), every pass under its own timeout.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
W=$1; TAG=$2
mkdir -p gpurun_out/$TAG
run() { timeout -k 10 150 rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/$TAG/$1 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-api --no-variants $BENCH_EXTRA --workload $W > gpurun_out/$TAG/$1.log 2>&1 || { echo "pass $1 failed"; return 1; }; echo "pass $1 ok"; }
run TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum || exit 1
run TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_RFIFO_STALL_CYCLES_sum TCP_GATE_EN2_sum || exit 1
run TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TD_TC_STALL_sum TD_SPI_STALL_sum || exit 1
run TCC_BUSY_sum TCC_TAG_STALL_sum TCC_REQ_sum TCC_READ_sum || exit 1
run SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_BUSY_CU_CYCLES || exit 1
python3 tools/pmc_summary.py gpurun_out/$TAG > /dev/null
