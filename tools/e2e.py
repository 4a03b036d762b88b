import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/viewport-entropy-toolkit_amd')
import numpy as np
import viewport_entropy_toolkit as vt
from viewport_entropy_toolkit.config import AnalyzerConfig
import bench
U, T = 1024, 30000
mu, mv = bench.synth_video(U, T, 1234, 0)
an = vt.SpatialEntropyAnalyzer(AnalyzerConfig(tile_counts=[500], output_dir='/tmp/vet_e2e'))
an.load_arrays(np.arange(T) * 0.1, mu, mv)
for i in range(3):
    t0 = time.perf_counter(); df = an.compute_entropy(); dt = time.perf_counter() - t0
    print('compute_entropy end-to-end (host arrays -> DataFrame): %.1f ms' % (dt * 1e3))
plan = an._get_plan()
for kw in (dict(want_assign=True, want_weights=True), dict(want_assign=True, want_weights=False), dict(want_assign=False, want_weights=False)):
    t0 = time.perf_counter(); plan.spatial(mu=mu, mv=mv, **kw); dt = time.perf_counter() - t0
    print(kw, '%.1f ms' % (dt * 1e3))
