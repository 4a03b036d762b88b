#!/bin/bash
# quick tuning sweep of the table-gather kernel on config3 (knobs read at table build / launch)
for il in 0 1; do for gs in 3 4 5; do for th in 256 512; do
  r=$(VET_TAB_INTERLEAVE=$il VET_GS_LOG2=$gs VET_LUT_THREADS=$th timeout -k 10 120 python bench.py --steps 8 --warmup 2 --workload config3 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3))")
  echo "interleave=$il GS=2^$gs threads=$th ms=$r"
done; done; done
