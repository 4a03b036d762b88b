#!/bin/bash
# usage: tools/collect_profiles.sh TAG RNN   (in the build container, after gpurun merged gpurun_out/TAG back)
# Copies what tools/refresh_profiles.sh produced into profiles/RNN/ and the two hash-guarded records bench.py reads
# (profiles/pmc_traffic.json, profiles/pmc_sq.json); prints the hash of the local kernel sources beside the records'.
cd "$(dirname "$0")/.."
TAG=$1; R=$2
mkdir -p profiles/$R
for f in gpurun_out/$TAG/bench_*.json gpurun_out/$TAG/default_bench_kernel_stats.csv gpurun_out/$TAG/default_bench_under_rocprof.json \
         gpurun_out/$TAG/pmc_config*_kernel_stats.csv gpurun_out/$TAG/formulation_timing.txt gpurun_out/$TAG/transition_any_timing.txt \
         gpurun_out/$TAG/transition_any_kernel_stats.csv gpurun_out/$TAG/*_pmc_sq.json gpurun_out/$TAG/*_pmc_stalls.json \
         gpurun_out/$TAG/weights_pass_timing.txt; do
  [ -f "$f" ] && cp "$f" profiles/$R/
done
cp gpurun_out/$TAG/pmc/pmc_traffic.json profiles/pmc_traffic.json && cp gpurun_out/$TAG/pmc/pmc_traffic.json profiles/$R/pmc_traffic.json
cp gpurun_out/$TAG/pmc_sq.json profiles/pmc_sq.json && cp gpurun_out/$TAG/pmc_sq.json profiles/$R/pmc_sq.json
python3 - <<'PY'
import json, sys
sys.path.insert(0, "tools")
from pmc_traffic import src_sha
t = json.load(open("profiles/pmc_traffic.json")); q = json.load(open("profiles/pmc_sq.json"))
print("kernel sources", src_sha(), "| pmc_traffic", t["config3"]["kernel_src_sha"], "| pmc_sq", q["config5"]["kernel_src_sha"])
PY
