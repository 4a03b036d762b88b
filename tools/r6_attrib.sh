#!/bin/bash
# round 6, VERDICT item 4: where config 4's time goes on the r05/r06 sources (stage cycles of the DEV build, the launch's
# fixed cost against the frame count, frames per workgroup)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
O=gpurun_out/r6_attrib; mkdir -p $O
B="--no-cpu-baseline --no-api --no-variants"
for w in config4 config2 defaults config3; do
  echo "== DEV build stage cycles: $w" >> $O/stage_cycles.txt
  VET_HIP_LIBRARY=$R/viewport-entropy-toolkit_amd/lib/dev/libvet_hip.so timeout -k 10 300 python3 bench.py --workload $w --steps 2 --warmup 1 $B 2>&1 >/dev/null | grep "k_spatial_lut" | tail -2 >> $O/stage_cycles.txt
done
echo "== tail probe (product build)" > $O/tail_probe.txt
timeout -k 10 300 python3 tools/tail_probe.py >> $O/tail_probe.txt 2>&1
for f in 1 2 4; do
  echo "== VET_LUT_FPW=$f" >> $O/tail_probe.txt
  VET_LUT_FPW=$f timeout -k 10 300 python3 tools/tail_probe.py >> $O/tail_probe.txt 2>&1
done
echo done
