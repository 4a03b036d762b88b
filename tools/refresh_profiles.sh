#!/bin/bash
# usage: tools/refresh_profiles.sh TAG [a|b]   (on the GPU box, from the repo root; a = traffic passes, bench lines and traces,
# b = the counter passes (pmc3 / pmc_sq), their records and the bench lines that read them: two gpurun calls of <= 20 minutes)
# Everything profiles/rNN/ holds about the current kernel sources, in one call: HBM traffic passes
# (pmc_traffic.sh), one bench line per workload / sample distribution, and the default bench under
# rocprofv3 --kernel-trace --stats.  Results land in gpurun_out/TAG/; copy them to profiles/rNN/.
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
TAG=$1; PART=${2:-ab}
mkdir -p gpurun_out/$TAG
line() { # name, args...
  local name=$1; shift
  timeout -k 10 300 python3 bench.py "$@" > gpurun_out/$TAG/bench_$name.json 2> gpurun_out/$TAG/bench_$name.err || { echo "bench $name failed"; tail -3 gpurun_out/$TAG/bench_$name.err; return 1; }
  echo "bench $name ok"
}
if [[ $PART == *a* ]]; then
bash tools/pmc_traffic.sh $TAG/pmc config3 config3u config4 config5 > gpurun_out/$TAG/pmc_traffic.log 2>&1 || { echo "pmc_traffic failed"; tail -5 gpurun_out/$TAG/pmc_traffic.log; exit 1; }
cp gpurun_out/$TAG/pmc/pmc_traffic.json profiles/pmc_traffic.json      # bench.py reads it from there
line config3 --workload config3 || exit 1
for d in clustered uniform; do line config3_$d --workload config3 --data $d --no-cpu-baseline || exit 1; done
line config3u --workload config3u --no-cpu-baseline || exit 1
line config5 --workload config5 --steps 50 --no-cpu-baseline || exit 1
for d in clustered uniform; do line config5_$d --workload config5 --steps 50 --data $d --no-cpu-baseline || exit 1; done
line config4 --workload config4 --no-cpu-baseline || exit 1
line config2 --workload config2 --steps 50 --no-cpu-baseline || exit 1
line config2x64 --workload config2x64 --no-cpu-baseline || exit 1
line defaults --workload defaults --no-cpu-baseline || exit 1
# batched launches of the nearest-tile and transition modes, and the same videos one call each
for w in config2x64u config2x64t config5x64; do
  line $w --workload $w --no-cpu-baseline || exit 1
  line ${w}_loop --workload $w --loop --no-cpu-baseline || exit 1
done
line config2x64_loop --workload config2x64 --loop --no-cpu-baseline || exit 1
# per-lattice tables instead of the fused one (VET_NO_FUSED=1)
for w in config4 defaults config2x64; do VET_NO_FUSED=1 line ${w}_per_lattice_tables --workload $w --no-cpu-baseline --no-api || exit 1; done
timeout -k 10 300 python3 tools/precise_timing.py > gpurun_out/$TAG/formulation_timing.txt 2>&1 || echo "formulation timing failed"
timeout -k 10 300 python3 tools/transition_any_timing.py > gpurun_out/$TAG/transition_any_timing.txt 2>&1 || echo "transition_any timing failed"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/transition_any_trace -- python3 tools/transition_any_timing.py > /dev/null 2>&1 && cp $(ls gpurun_out/$TAG/transition_any_trace/*/*_kernel_stats.csv | head -1) gpurun_out/$TAG/transition_any_kernel_stats.csv
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$TAG/default_trace -- python3 bench.py --no-cpu-baseline --no-api --no-variants > gpurun_out/$TAG/default_bench_under_rocprof.json 2> gpurun_out/$TAG/default_trace.err || { echo "rocprof default bench failed"; exit 1; }
cp $(ls gpurun_out/$TAG/default_trace/*/*_kernel_stats.csv | head -1) gpurun_out/$TAG/default_bench_kernel_stats.csv
for w in config3 config3u config4 config5; do cp $(ls gpurun_out/$TAG/pmc/$w/trace/*/*_kernel_stats.csv | head -1) gpurun_out/$TAG/pmc_${w}_kernel_stats.csv; done
fi
if [[ $PART == *b* ]]; then
# L2 / L1 counters of the table kernel at config 3 and SQ instruction mix (config 3 and 4): pmc3.sh, pmc_sq.sh
bash tools/pmc3.sh config3 $TAG/pmc_stalls_config3 > gpurun_out/$TAG/pmc3_config3.log 2>&1 && cp gpurun_out/$TAG/pmc_stalls_config3/summary.json gpurun_out/$TAG/config3_pmc_stalls.json
for w in config3 config4 config5 defaults; do bash tools/pmc_sq.sh $w $TAG/pmc_sq_$w > gpurun_out/$TAG/pmc_sq_$w.log 2>&1 && cp gpurun_out/$TAG/pmc_sq_$w/summary.json gpurun_out/$TAG/${w}_pmc_sq.json; done
# L2 requests of the fused kernels (config 4, reference defaults)
for w in config4 defaults; do bash tools/pmc3.sh $w $TAG/pmc_stalls_$w > gpurun_out/$TAG/pmc3_$w.log 2>&1 && cp gpurun_out/$TAG/pmc_stalls_$w/summary.json gpurun_out/$TAG/${w}_pmc_stalls.json; done
# SQ record bench.py's transition `secondary` block reads (sha-guarded like pmc_traffic.json)
python3 tools/pmc_sq_record.py config5=gpurun_out/$TAG/config5_pmc_sq.json config4=gpurun_out/$TAG/config4_pmc_sq.json+gpurun_out/$TAG/config4_pmc_stalls.json config3=gpurun_out/$TAG/config3_pmc_sq.json+gpurun_out/$TAG/config3_pmc_stalls.json defaults=gpurun_out/$TAG/defaults_pmc_sq.json+gpurun_out/$TAG/defaults_pmc_stalls.json > /dev/null && cp profiles/pmc_sq.json gpurun_out/$TAG/pmc_sq.json
line config5_with_sq --workload config5 --steps 50 --no-cpu-baseline --no-api
timeout -k 10 300 python3 tools/weights_pass_timing.py > gpurun_out/$TAG/weights_pass_timing.txt 2>&1 || echo "weights pass timing failed"
fi
echo "refresh $PART done"
