#!/bin/bash
# needs tools/experiments/r6_grid_stride_atomic_prefetch.patch applied (VET_LUT_PERSIST is not a knob of the product)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
O=gpurun_out/r6_attrib; mkdir -p $O
python3 -m pytest tests/test_hip_shapes.py -x -q -k "fused_table_kernels_agree" > $O/persist2_tests.log 2>&1; tail -3 $O/persist2_tests.log
echo "== grid-stride form with atomic block claims and the next block's samples requested ahead" > $O/persist2_ab.txt
bash tools/ab_env.sh "VET_LUT_PERSIST=0 VET_LUT_PERSIST=8 VET_LUT_PERSIST=7,VET_LUT_OCC8=0 VET_LUT_PERSIST=6,VET_LUT_OCC8=0" "config4 defaults" 2 >> $O/persist2_ab.txt 2>&1
cat $O/persist2_ab.txt
