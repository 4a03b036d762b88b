#!/bin/bash
# usage: tools/pmc.sh <workload> <tag>   — SQ/TCC counter passes for the dominant kernel
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R
W=$1; TAG=$2
mkdir -p gpurun_out/$TAG
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/$TAG/sq1 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload $W > gpurun_out/$TAG/sq1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM --output-format csv -d $R/gpurun_out/$TAG/sq2 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload $W > gpurun_out/$TAG/sq2.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/$TAG/fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload $W > gpurun_out/$TAG/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/$TAG/write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --workload $W > gpurun_out/$TAG/write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/trace -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --workload $W > gpurun_out/$TAG/trace.log 2>&1
python3 tools/pmc_summary.py gpurun_out/$TAG
