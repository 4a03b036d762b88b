#!/bin/bash
# usage: tools/pmc_traffic.sh TAG WORKLOAD...      (on the GPU box, from the repo root)
# HBM traffic of the dominant kernel of each workload: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc
# passes (never combined with a trace domain), then a --kernel-trace --stats pass for the kernel time, folded by
# tools/pmc_traffic.py into gpurun_out/TAG/pmc_traffic.json together with the hash of the kernel sources the
# numbers were measured on (bench.py drops a record whose hash differs from the sources it runs on).
# Every pass runs under its own `timeout -k 10 150` (a hung pass must not take the call's whole budget).
# Copy that file to profiles/pmc_traffic.json and the per-workload summaries to profiles/rNN/.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
TAG=$1; shift
mkdir -p gpurun_out/$TAG
for W in "$@"; do
  for C in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 150 rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/$TAG/$W/$C -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-api --no-variants $BENCH_EXTRA --workload $W > gpurun_out/$TAG/$W.$C.log 2>&1
  done
  timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/$W/trace -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-api --no-variants $BENCH_EXTRA --workload $W > gpurun_out/$TAG/$W.trace.log 2>&1
  echo "pmc passes of $W done"
done
python3 tools/pmc_traffic.py gpurun_out/$TAG "$@"
