#!/bin/bash
# quick tuning sweep of the table-gather kernel on config3
for un in 2 4 8; do for gs in 4 5 6; do for th in 256 512 1024; do
  r=$(VET_UN=$un VET_GS_LOG2=$gs VET_LUT_THREADS=$th timeout -k 10 120 python bench.py --steps 8 --warmup 2 --workload config3 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3))")
  echo "UN=$un GS=2^$gs threads=$th ms=$r"
done; done; done
