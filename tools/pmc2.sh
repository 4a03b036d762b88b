#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
W=$1; TAG=$2
mkdir -p gpurun_out/$TAG
run() { rocprofv3 --pmc "$@" --output-format csv -d $R/gpurun_out/$TAG/$1 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload $W > gpurun_out/$TAG/$1.log 2>&1; }
run SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU
run SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM
run TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum
run TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
run TA_BUSY_avr TA_TA_BUSY_sum TD_TD_BUSY_sum
python3 tools/pmc_summary.py gpurun_out/$TAG
