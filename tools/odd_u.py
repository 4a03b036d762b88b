import sys
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/viewport-entropy-toolkit_amd')
import numpy as np, torch
from viewport_entropy_toolkit import _native, _quantiser
import bench
eng = _native.Engine(0)
dev = torch.device('cuda', 0)
for U in (1024, 1023, 59):
    T = 30000
    mu_h, mv_h = bench.synth_video(U, T, 1, 0)
    mu = torch.from_numpy(mu_h).to(dev); mv = torch.from_numpy(mv_h).to(dev)
    ent = torch.empty(T, dtype=torch.float64, device=dev); idx = torch.empty((T, U), dtype=torch.int32, device=dev)
    plan = _native.Plan(eng, [_quantiser.lattice_xyz(500)], 120.0, 2.0, False, 100, 200)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3): plan.spatial_device(mu.data_ptr(), mv.data_ptr(), U, T, ent.data_ptr(), d_assign=idx.data_ptr(), stream=st)
    torch.cuda.synchronize(); eng.profile_enable(True); eng.profile_reset()
    for _ in range(20): plan.spatial_device(mu.data_ptr(), mv.data_ptr(), U, T, ent.data_ptr(), d_assign=idx.data_ptr(), stream=st)
    torch.cuda.synchronize()
    ms, n = eng.profile_get('k_spatial'); eng.profile_enable(False)
    print(f'U={U}: {ms/n:.4f} ms/launch  {20*U*T/(ms/n*1e-3)/1e9:.0f} GB/s algorithmic')
    plan.close()
