import sys, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/viewport-entropy-toolkit_amd')
import numpy as np
from viewport_entropy_toolkit import _native
from oracle import vet_oracle as vo
eng = _native.Engine(0)
W, H = 3840, 1920
rng = np.random.default_rng(0)
mu, mv = rng.random((6, 40)), rng.random((6, 40))
t0 = time.perf_counter()
plan = _native.Plan(eng, [vo.fibonacci_lattice(500)], 120.0, 2.0, True, W, H)
plan.set_table_policy(1)
t1 = time.perf_counter()
res = plan.spatial(mu=mu, mv=mv)
t2 = time.perf_counter()
res2 = plan.spatial(mu=mu, mv=mv)
t3 = time.perf_counter()
print('plan %.1f ms, first call (table build: %d rows for %d directions, stride %d, %.1f GB) %.1f ms, second call %.2f ms' % (
    (t1 - t0) * 1e3, plan.table_rows(), plan.n_dirs, plan.table_stride(0), (plan.table_rows() + 1) * plan.table_stride(0) * 6 / 1e9, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
ent, assign, _ = vo.spatial_series(mu, mv, W, H, [500])
assert np.array_equal(res['assign'], assign)
np.testing.assert_allclose(res['entropy'], ent, rtol=1e-8)
print('parity ok on a 3840x1920 grid through the weight table')
