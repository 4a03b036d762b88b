"""Transition kernel time against the number of rows (512 users, 200 tiles): separates the per-launch cost
from the per-row cost.  usage: python tools/transition_scaling.py [with_pairs]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'viewport-entropy-toolkit_amd'))
import numpy as np, torch
from viewport_entropy_toolkit import _native, _quantiser
import bench
U = 512
dev = torch.device('cuda', 0)
eng = _native.Engine(0)
plan = _native.Plan(eng, [_quantiser.lattice_xyz(200)], 120.0, 2.0, True, 100, 200)
with_pairs = len(sys.argv) > 1
for T in (258, 1025, 2049, 4097, 10000, 20481, 40961):
    mu_h, mv_h = bench.synth_video(U, T, 1234, 0)
    mu = torch.from_numpy(mu_h).to(dev); mv = torch.from_numpy(mv_h).to(dev)
    ent = torch.empty(T, dtype=torch.float64, device=dev); idx = torch.empty((T, U, 2), dtype=torch.int32, device=dev)
    st = torch.zeros(2, dtype=torch.int32, device=dev)
    run_stream = torch.cuda.Stream(device=dev)
    def step():
        plan.transition_device(mu.data_ptr(), mv.data_ptr(), U, T, ent.data_ptr(), d_pairs=idx.data_ptr() if with_pairs else None,
                               d_status=st.data_ptr(), stream=run_stream.cuda_stream)
    for _ in range(5): step()
    torch.cuda.synchronize()
    eng.profile_enable(True); eng.profile_reset()
    for _ in range(30): step()
    torch.cuda.synchronize()
    ms, n = eng.profile_get('k_transition')
    eng.profile_enable(False)
    print(f"T={T:6d} rows/wg={(T - 1) / 2048:6.2f} kernel {ms / n * 1e3:8.2f} us  pairs={with_pairs}", flush=True)
