import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/viewport-entropy-toolkit_amd')
import numpy as np
from viewport_entropy_toolkit import _native, _quantiser
import bench
eng = _native.Engine(0)
for name in ('config2',):
    U, T, tcs, mode, weighted = bench.WORKLOADS[name]
    for (u, t) in ((8, 300), (U, T), (256, 10000)):
        mu, mv = bench.synth_video(u, t, 1, 0)
        plan = _native.Plan(eng, [_quantiser.lattice_xyz(tc) for tc in tcs], 120.0, 2.0, True, 100, 200)
        for _ in range(15): plan.spatial(mu=mu, mv=mv)
        n = 50
        t0 = time.perf_counter()
        for _ in range(n): plan.spatial(mu=mu, mv=mv)
        dt = (time.perf_counter() - t0) / n
        print(f'host-path call latency {u} users x {t} frames, {tcs}: {dt*1e3:.3f} ms per call')
        plan.close()
