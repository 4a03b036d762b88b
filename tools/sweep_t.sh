#!/bin/bash
for rpw in 1 2 4 8; do for th in 256 512; do
  r=$(VET_T_RPW=$rpw VET_T_THREADS=$th timeout -k 10 120 python bench.py --steps 30 --warmup 3 --workload config5 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['roofline']['achieved'],1))")
  echo "rpw=$rpw threads=$th ms, GB/s = $r"
done; done
