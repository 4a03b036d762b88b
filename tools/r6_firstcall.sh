#!/bin/bash
# first-call cost of the table formulation against the grid (the alias table is built on the device since round 6)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
O=gpurun_out/r6_firstcall; mkdir -p $O
for spec in "config3 100x200 1" "config2 100x200 1" "config3 200x400 1" "config2 200x400 1" "config3 3840x1920 1" "config2 200x400 -1" "config3 3840x1920 -1"; do
  set -- $spec
  timeout -k 10 600 python3 bench.py --no-cpu-baseline --no-api --no-variants --steps 5 --warmup 1 --workload $1 --grid $2 --policy $3 > $O/fc_$1_$2_p$3.json 2> $O/fc.err || { echo "$spec FAILED"; tail -3 $O/fc.err; continue; }
  python3 - "$spec" $O/fc_$1_$2_p$3.json <<'PY'
import json, sys
j = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
print(f"{sys.argv[1]:28s} plan_build {j['plan_build_ms']:8.2f} ms  first_call {j['first_call_ms']:9.2f} ms  table_build(kernels) {j['table_build_ms']:7.2f} ms  step {j['ms_per_step']:.4f} ms  rows {(j.get('table') or {}).get('rows')}  parity {j['parity']['ok']}")
PY
done | tee $O/first_call.txt
