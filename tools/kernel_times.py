import sys, os, json
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/viewport-entropy-toolkit_amd')
import numpy as np, torch
from viewport_entropy_toolkit import _native, _quantiser
import bench
U, T, tcs, mode, weighted = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else 'config3']
mu_h, mv_h = bench.synth_video(U, T, 1234, 0)
dev = torch.device('cuda', 0)
mu = torch.from_numpy(mu_h).to(dev); mv = torch.from_numpy(mv_h).to(dev)
ent = torch.empty(T, dtype=torch.float64, device=dev); idx = torch.empty((T, U), dtype=torch.int32, device=dev)
st = torch.zeros(2, dtype=torch.int32, device=dev)
eng = _native.Engine(0)
plan = _native.Plan(eng, [_quantiser.lattice_xyz(tc) for tc in tcs], 120.0, 2.0, weighted, 100, 200)
stream = torch.cuda.current_stream().cuda_stream
def step():
    plan.spatial_device(mu.data_ptr(), mv.data_ptr(), U, T, ent.data_ptr(), d_assign=idx.data_ptr(), d_status=st.data_ptr(), stream=stream)
for _ in range(3): step()
torch.cuda.synchronize()
eng.profile_enable(True); eng.profile_reset()
N = 10
for _ in range(N): step()
torch.cuda.synchronize()
for k in _native.KERNEL_IDS:
    ms, n = eng.profile_get(k)
    if n: print(k, 'launches/step', n / N, 'ms/launch %.4f' % (ms / n))
