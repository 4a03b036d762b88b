#!/bin/bash
# prof_bench.sh NAME [bench.py args...] — one bench run under rocprofv3 --kernel-trace --stats; prints the kernel table
# (run on the GPU box: gpurun -- tools/prof_bench.sh config3 --workload config3)
name=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$name
rm -rf $out && mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-api --steps 10 "$@" > $out/bench.json 2> $out/bench.err
f=$(find $out -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:8]:
    print(f"{r['Name'][:90]:90s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs'])/1e3:10.1f} pct {r['Percentage']}")
PY
cp "$f" $GRAFT_REPO_ROOT/gpurun_out/prof_${name}_kernel_stats.csv
