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
for n in (1, 256, 256, 4096, 256):
    for which in (0, 1):
        t0 = time.perf_counter(); a = res.rows(which, 1000, n); dt = time.perf_counter() - t0
        print("which", which, "rows", n, "bytes", a.nbytes, "ms", round(dt * 1e3, 3))
from viewport_entropy_toolkit._results import DeviceRows, TileAssignments
names = [f"u{i}" for i in range(1024)]
dr = DeviceRows(res, 0, 30000)
for i in (0, 1, 300, 5000, 5001):
    t0 = time.perf_counter(); row = dr[i]; t1 = time.perf_counter(); d = dict(TileAssignments(names, row)); t2 = time.perf_counter()
    print(i, "fetch ms", round((t1 - t0) * 1e3, 3), "dict ms", round((t2 - t1) * 1e3, 3))
