#!/usr/bin/env python3
"""kt.py [bench.py args] — one short bench run, prints step / per-kernel times (engine hipEvents)."""
import json, subprocess, sys, os
r = subprocess.run([sys.executable, os.path.join(os.path.dirname(__file__), "..", "bench.py"), "--no-cpu-baseline", "--no-api",
                    "--no-variants", "--steps", "10"] + sys.argv[1:], capture_output=True, text=True)
if r.returncode:
    print(r.stderr[-2000:]); sys.exit(1)
d = json.loads(r.stdout.strip().splitlines()[-1])
print(f"step {d['ms_per_step']:.4f} ms  kernels {d['kernel_ms_per_step']}")
