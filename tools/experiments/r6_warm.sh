#!/bin/bash
# round 6: is the first wave of workgroups slow because the table is not in the XCDs' L2 when a launch starts? (VET_LUT_WARM, same box)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
O=gpurun_out/r6_attrib; mkdir -p $O
echo "== warm A/B" > $O/warm_ab.txt
bash tools/ab_env.sh "VET_LUT_WARM=0 VET_LUT_WARM=100 VET_LUT_WARM=60 VET_LUT_WARM=30" "config4 defaults config2" 3 >> $O/warm_ab.txt 2>&1
cat $O/warm_ab.txt
