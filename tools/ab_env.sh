#!/bin/bash
# ab_env.sh — same-box A/B of environment knobs on ONE build (the knobs are read once per context, each run is a process
# of its own).   usage (on the GPU box): tools/ab_env.sh "VAR=a VAR=b ..." "workload ..." [reps]
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
reps=${3:-2}
for w in $2; do
  for r in $(seq $reps); do
    for kv in $1; do
      printf "%-10s %-22s " "$w" "$kv"
      env ${kv//,/ } python3 tools/kt.py --workload $w     # VAR=a,VAR2=b sets several
    done
  done
done
