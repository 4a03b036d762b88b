#!/bin/bash
# usage: tools/grid_sensitivity.sh TAG     (on the GPU box)  VERDICT r05 item 5: throughput against the pixel grid.
# config 3 and config 2 on the reference's default 100x200 grid, its README example 200x400 (README.md:91-92) and a
# 3840x1920 grid, under policy 0 (table iff >= 8 samples per direction), +1 (table) and -1 (sweep); for the 200x400
# runs the L2 request / HBM fetch counters.  Output: gpurun_out/TAG/grid_sensitivity.txt (+ the bench lines).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"
TAG=$1; O=gpurun_out/$TAG; mkdir -p $O
row() { # name, bench args...
  local name=$1; shift
  timeout -k 10 600 python3 bench.py --no-cpu-baseline --no-api --no-variants --steps 10 --warmup 2 "$@" > $O/grid_$name.json 2> $O/grid_$name.err || { echo "$name FAILED: $(tail -2 $O/grid_$name.err | tr '\n' ' ')" >> $O/grid_sensitivity.txt; return 0; }
  python3 - "$name" "$O/grid_$name.json" >> $O/grid_sensitivity.txt <<'PY'
import json, sys
name, f = sys.argv[1:3]
j = json.loads(open(f).read().strip().splitlines()[-1])
r = j["roofline"]; s = r.get("secondary") or {}; t = j.get("table") or {}
print(f"{name:34s} grid {j['config']['grid'][0]}x{j['config']['grid'][1]}  formulation {(j.get('formulation_bound') or {}).get('formulation', '-'):7s}"
      f" kernel {r['avg_kernel_ms']:.4f} ms  step {j['ms_per_step']:.4f} ms  samples/s {j['value']:.3g}  frac {r['frac']:.4f}"
      f"  table {t.get('bytes', 0) / 1e6:.1f} MB ({t.get('rows', 0)} rows x {t.get('stride', 0)})  table_build {j['table_build_ms']:.2f} ms"
      f"  first_call {j['first_call_ms']:.1f} ms  distinct/frame {s.get('distinct_directions_per_frame', float('nan')):.1f}"
      f"  parity {j['parity']['ok']} (rel {j['parity']['entropy_max_rel']:.1e})")
PY
}
echo "# pixel-grid sensitivity ($(date -u +%F)), kernel sources $(python3 -c 'import bench; print(bench.kernel_src_sha())')" > $O/grid_sensitivity.txt
for g in 100x200 200x400; do
  for w in config3 config2; do
    for p in 0 1 -1; do row ${w}_${g}_policy$p --workload $w --grid $g --policy $p; done
  done
done
for p in 0 1; do row config3_3840x1920_policy$p --workload config3 --grid 3840x1920 --policy $p; done
row config4_200x400_policy0 --workload config4 --grid 200x400
# counters of the 200x400 table run: L2 requests, HBM-side fetch
for C in TCC_REQ_sum FETCH_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $C --output-format csv -d $R/$O/pmc_200x400/$C -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-api --no-variants --workload config3 --grid 200x400 --policy 1 > $O/pmc_200x400_$C.log 2>&1
done
python3 - $O >> $O/grid_sensitivity.txt <<'PY'
import csv, glob, sys, collections
root = sys.argv[1]
for c in ("TCC_REQ_sum", "FETCH_SIZE"):
    vals = collections.defaultdict(list)
    for f in glob.glob(f"{root}/pmc_200x400/{c}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "k_spatial_lut" in r["Kernel_Name"]:
                vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in vals.items():
        note = " (x 128 B = %.2f GB)" % (sum(v) / len(v) * 128 / 1e9) if "TCC_REQ" in k else " (KB x 2 for gfx950 = %.2f GB)" % (sum(v) / len(v) * 2 * 1024 / 1e9)
        print(f"config3 200x400 policy +1, k_spatial_lut: {k} = {sum(v) / len(v):.4g} per launch{note}")
PY
cat $O/grid_sensitivity.txt
