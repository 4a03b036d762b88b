#!/bin/bash
# round 6: does starting the first wave of workgroups out of phase shorten a short launch? (VET_LUT_STAGGER, same box)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
O=gpurun_out/r6_attrib; mkdir -p $O
echo "== stagger A/B" > $O/stagger_ab.txt
bash tools/ab_env.sh "VET_LUT_STAGGER=0 VET_LUT_STAGGER=1 VET_LUT_STAGGER=2 VET_LUT_STAGGER=3 VET_LUT_STAGGER=5 VET_LUT_STAGGER=8" "config4 defaults config3 config2" 2 >> $O/stagger_ab.txt 2>&1
cat $O/stagger_ab.txt
