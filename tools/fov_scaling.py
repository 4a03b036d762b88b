"""config 3 (1024 users x 30 000 frames, 501 tiles) at several FoV angles: the rows of the weight table get shorter while
the number of row walks stays the same, which separates the per-row from the per-entry cost of k_spatial_lut."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'viewport-entropy-toolkit_amd'))
import numpy as np, torch
from viewport_entropy_toolkit import _native, _quantiser
import bench
U, T = 1024, 30000
dev = torch.device('cuda', 0)
mu_h, mv_h = bench.synth_video(U, T, 1234, 0, sys.argv[1] if len(sys.argv) > 1 else 'random_walk')
mu = torch.from_numpy(mu_h).to(dev); mv = torch.from_numpy(mv_h).to(dev)
ent = torch.empty(T, dtype=torch.float64, device=dev); idx = torch.empty((T, U), dtype=torch.int32, device=dev)
st = torch.zeros(2, dtype=torch.int32, device=dev)
eng = _native.Engine(0)
run_stream = torch.cuda.Stream(device=dev)
for fov in (120.0, 100.0, 90.0, 75.0, 60.0, 40.0, 20.0):
    plan = _native.Plan(eng, [_quantiser.lattice_xyz(500)], fov, 2.0, True, 100, 200)
    plan.set_table_policy(1)
    def step():
        plan.spatial_device(mu.data_ptr(), mv.data_ptr(), U, T, ent.data_ptr(), d_assign=idx.data_ptr(), d_status=st.data_ptr(),
                            stream=run_stream.cuda_stream)
    for _ in range(3): step()
    torch.cuda.synchronize()
    eng.profile_enable(True); eng.profile_reset()
    for _ in range(10): step()
    torch.cuda.synchronize()
    ms, n = eng.profile_get('k_spatial')
    eng.profile_enable(False)
    cap = (1 - np.cos(np.radians(fov / 2))) / 2 * 501
    print(f"fov {fov:5.0f}  entries/row ~{cap:6.1f}  stride {plan.table_stride(0):4d}  form {plan.last_formulation(0):7s}  kernel {ms / n:7.4f} ms", flush=True)
    plan.close()
