#!/bin/bash
run() { r=$(env "$@" timeout -k 10 120 python bench.py --steps 10 --warmup 2 --workload config3 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3))"); echo "$* -> $r"; }
for gs in 3 4 5; do for un in 2 4 8; do run VET_PART=0 VET_GS_LOG2=$gs VET_UN=$un; done; done
for gs in 3 4 5; do run VET_PART=1 VET_GS_LOG2=$gs VET_PART_THREADS=128; run VET_PART=1 VET_GS_LOG2=$gs VET_PART_THREADS=256; done
