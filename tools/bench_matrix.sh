#!/bin/bash
# tools/bench_matrix.sh OUT [ENV=VAL ...] -- runs bench.py config3 over the sample distributions, prints ms
out=$1; shift
mkdir -p gpurun_out
for d in ${DISTS:-random_walk clustered uniform single fixed}; do
  env "$@" python bench.py --data $d --no-cpu-baseline --steps ${STEPS:-10} --workload ${WL:-config3} > gpurun_out/${out}_$d.json 2> gpurun_out/${out}_$d.err || tail -3 gpurun_out/${out}_$d.err
  python - <<PY
import json
try:
    j=json.loads(open("gpurun_out/${out}_$d.json").read().strip().splitlines()[-1])
    s=j["roofline"].get("secondary",{})
    print("${out}", "$d", "ms", round(j["roofline"]["avg_kernel_ms"],4), "distinct", s.get("distinct_directions_per_frame"), "sol", {k:round(v,3) for k,v in (s.get("sol_ms") or {}).items()})
except Exception as e: print("${out}", "$d", "ERR", e)
PY
done
