#!/bin/bash
# round 6 experiment: phased launches of the integer table kernel (VET_LUT_PHASES), same box
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
O=gpurun_out/r6_phases; mkdir -p $O
python3 -m pytest tests/test_hip_shapes.py -x -q -k "phased_table" > $O/tests.log 2>&1; tail -4 $O/tests.log
echo "== phases A/B" > $O/phases_ab.txt
bash tools/ab_env.sh "VET_LUT_PHASES=1 VET_LUT_PHASES=2 VET_LUT_PHASES=4 VET_LUT_PHASES=8" "config3 config4 defaults" 2 >> $O/phases_ab.txt 2>&1
cat $O/phases_ab.txt
