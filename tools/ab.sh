#!/bin/bash
# ab.sh — same-box A/B of two builds of libvet_hip.so (the pool's boxes differ by ~5 %, so variants are only
# comparable inside one gpurun call).   usage (on the GPU box): tools/ab.sh "libA libB ..." "workload ..." [reps]
# a lib name is a directory under viewport-entropy-toolkit_amd/ holding libvet_hip.so (lib = the product build)
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
reps=${3:-2}
for w in $2; do
  for r in $(seq $reps); do
    for l in $1; do
      printf "%-10s %-12s " "$w" "$l"
      VET_HIP_LIBRARY=$R/viewport-entropy-toolkit_amd/$l/libvet_hip.so python3 tools/kt.py --workload $w
    done
  done
done
