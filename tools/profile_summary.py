"""One line per bench record of a refresh_profiles.sh run: python tools/profile_summary.py gpurun_out/TAG"""
import glob, json, os, sys
root = sys.argv[1]
for f in sorted(glob.glob(root + '/bench_*.json')) + [root + '/default_bench_under_rocprof.json']:
    try:
        j = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:  # noqa: BLE001
        print(os.path.basename(f), 'ERR', e)
        continue
    r = j['roofline']
    print(f"{os.path.basename(f):36s} ms/step {j['ms_per_step']:.4f} kernel {r['avg_kernel_ms']:.4f} value {j['value']:.3g} "
          f"frac {(r['frac'] or 0):.3f} traffic {r['traffic']}")
