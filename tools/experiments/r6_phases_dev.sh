#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
O=gpurun_out/r6_phases; mkdir -p $O
B="--no-cpu-baseline --no-api --no-variants --steps 1 --warmup 1"
for w in config3 defaults config4; do
  for P in 1 2 4; do
    echo "== $w VET_LUT_PHASES=$P (DEV build, VET_FUSED=1: stage cycles per workgroup and launch)"
    VET_FUSED=1 VET_LUT_PHASES=$P VET_HIP_LIBRARY=$R/viewport-entropy-toolkit_amd/lib/dev/libvet_hip.so timeout -k 10 300 python3 bench.py --workload $w $B 2>&1 >/dev/null | grep "k_spatial_lut" | tail -$P
  done
done > $O/phases_stage_cycles.txt 2>&1
cat $O/phases_stage_cycles.txt
