#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
O=gpurun_out/r6_attrib; mkdir -p $O
B="--no-cpu-baseline --no-api --no-variants --steps 3 --warmup 2"
for w in config4; do
  for env in "VET_LUT_MIXED=0" "VET_LUT_MIXED=0 VET_LUT_FPW=1" "VET_LUT_MIXED=1"; do
    env $env VET_LUT_TIMELINE=$R/$O/tl.bin VET_HIP_LIBRARY=$R/viewport-entropy-toolkit_amd/lib/dev/libvet_hip.so timeout -k 10 300 python3 bench.py --workload $w $B > /dev/null 2> $O/timeline_$w.err
    echo "== $w $env"; python3 tools/timeline_summary.py $O/tl.bin
  done
done > $O/timeline.txt 2>&1
rm -f $O/tl.bin
cat $O/timeline.txt
