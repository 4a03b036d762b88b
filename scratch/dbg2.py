import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/viewport-entropy-toolkit_amd')
import numpy as np
np.set_printoptions(linewidth=200, precision=4, suppress=True)
from viewport_entropy_toolkit import _native, _synthetic
from oracle import vet_oracle as vo
eng = _native.Engine(0)
mu, mv = _synthetic.random_walk_video(8, 4, base_seed=1234)
for tcs in ([100],):
    plan = _native.Plan(eng, [vo.fibonacci_lattice(t) for t in tcs], 120.0, 2.0, True, 100, 200)
    res = plan.spatial(mu=mu, mv=mv, want_weights=True)
    ent, assign, w = vo.spatial_series(mu, mv, 100, 200, tcs, want_weights=True)
    print('gpu', res['weights'][0])
    print('ora', w[0])
    print('diff idx', np.nonzero(~np.isclose(res['weights'][0], w[0], atol=1e-9))[0])
    print(res['weights'][0].sum(), w[0].sum())
    plan.close()
