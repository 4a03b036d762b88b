"""The reference's default tile_counts [20, 50, 100, 250, 1000] at 256 users x 10 000 frames: kernel time of subsets of the
lattices, to see what the fused launch costs per lattice."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'viewport-entropy-toolkit_amd'))
import numpy as np, torch
from viewport_entropy_toolkit import _native, _quantiser
import bench
U, T = 256, 10000
dev = torch.device('cuda', 0)
mu_h, mv_h = bench.synth_video(U, T, 1234, 0)
mu = torch.from_numpy(mu_h).to(dev); mv = torch.from_numpy(mv_h).to(dev)
ent = torch.empty(T, dtype=torch.float64, device=dev); idx = torch.empty((T, U), dtype=torch.int32, device=dev)
st = torch.zeros(2, dtype=torch.int32, device=dev)
eng = _native.Engine(0)
s = torch.cuda.Stream(device=dev)
for tcs in ([20, 50, 100, 250, 1000], [1000], [250], [100], [50], [20], [20, 50, 100], [250, 1000], [20, 50, 100, 250]):
    plan = _native.Plan(eng, [_quantiser.lattice_xyz(tc) for tc in tcs], 120.0, 2.0, True, 100, 200)
    def step():
        plan.spatial_device(mu.data_ptr(), mv.data_ptr(), U, T, ent.data_ptr(), d_assign=idx.data_ptr(), d_status=st.data_ptr(), stream=s.cuda_stream)
    for _ in range(3): step()
    torch.cuda.synchronize()
    eng.profile_enable(True); eng.profile_reset()
    for _ in range(10): step()
    torch.cuda.synchronize()
    ms, n = eng.profile_get('k_spatial')
    eng.profile_enable(False)
    print(f"tile_counts {str(tcs):28s} launches/step {n / 10:.0f}  kernel ms/step {ms / 10:7.4f}  form {plan.last_formulation(0)}", flush=True)
    plan.close()
