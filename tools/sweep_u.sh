#!/bin/bash
for w in 1 2 3 4; do
  r=$(VET_U_WGS_PER_CU=$w timeout -k 10 120 python bench.py --steps 20 --warmup 3 --workload config3u --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), round(d['roofline']['achieved'],1))")
  echo "wgs/cu=$w ms, GB/s = $r"
done
