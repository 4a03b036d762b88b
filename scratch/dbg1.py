import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/viewport-entropy-toolkit_amd')
import numpy as np
from viewport_entropy_toolkit import _native, _synthetic
from oracle import vet_oracle as vo
eng = _native.Engine(0)
mu, mv = _synthetic.random_walk_video(8, 300, base_seed=1234)
for tcs in ([50], [100], [200], [20], [50, 100], [50,100,200]):
    plan = _native.Plan(eng, [vo.fibonacci_lattice(t) for t in tcs], 120.0, 2.0, True, 100, 200)
    res = plan.spatial(mu=mu, mv=mv)
    ent, assign, _ = vo.spatial_series(mu, mv, 100, 200, tcs)
    bad = np.nonzero(~np.isclose(res['entropy'], ent, rtol=1e-9))[0]
    print(tcs, 'bad frames', len(bad), bad[:10], res['entropy'][bad[:4]], ent[bad[:4]])
    plan.close()
