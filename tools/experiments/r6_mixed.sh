#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
O=gpurun_out/r6_attrib; mkdir -p $O
echo "== mixed A/B (big workgroups of the uniform rule's size, tail of half-size ones)" > $O/mixed_ab2.txt
bash tools/ab_env.sh "VET_LUT_MIXED=0 VET_LUT_MIXED=1" "config4" 4 >> $O/mixed_ab2.txt 2>&1
cat $O/mixed_ab2.txt
VET_LUT_TIMELINE=$R/$O/tl.bin VET_HIP_LIBRARY=$R/viewport-entropy-toolkit_amd/lib/dev/libvet_hip.so timeout -k 10 300 python3 bench.py --workload config4 --no-cpu-baseline --no-api --no-variants --steps 3 --warmup 2 > /dev/null 2>&1
python3 tools/timeline_summary.py $O/tl.bin | head -8 | tee $O/timeline_mixed2.txt; rm -f $O/tl.bin
