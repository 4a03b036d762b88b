#!/bin/bash
# needs tools/experiments/r6_grid_stride_stagger_warm.patch applied (VET_LUT_PERSIST is not a knob of the product)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
O=gpurun_out/r6_attrib; mkdir -p $O
# grid-stride form of the fused kernel (VET_LUT_PERSIST = resident workgroups per CU), same box
echo "== persist A/B" > $O/persist_ab.txt
bash tools/ab_env.sh "VET_LUT_PERSIST=0 VET_LUT_PERSIST=8 VET_LUT_PERSIST=7,VET_LUT_OCC8=0 VET_LUT_PERSIST=8,VET_LUT_FPW=1 VET_LUT_PERSIST=4" "config4 defaults" 2 >> $O/persist_ab.txt 2>&1
