import sys, time
sys.path.insert(0, 'viewport-entropy-toolkit_amd'); sys.path.insert(0, '.')
import numpy as np
from viewport_entropy_toolkit import _native, _quantiser
from bench import synth_video
mu, mv = synth_video(1024, 30000, 1234, 0)
eng = _native.Engine(0)
plan = _native.Plan(eng, [_quantiser.lattice_xyz(500)], 120.0, 2.0, True, 100, 200)
r = plan.spatial_resident(mu=mu, mv=mv)
res = r["result"]
for n in (256, 512, 1024, 2048, 4096, 8192, 4096, 2048, 1024, 256):
    ts = []
    for rep in range(3):
        t0 = time.perf_counter(); a = res.rows(1, 1000, n); ts.append((time.perf_counter() - t0) * 1e3)
    t0 = time.perf_counter(); b = res.rows(0, 1000, n); t_a = (time.perf_counter() - t0) * 1e3
    print(f"rows {n:5d}: weights {ts[0]:8.3f} {ts[1]:8.3f} {ts[2]:8.3f} ms   assignments {t_a:8.3f} ms", flush=True)
