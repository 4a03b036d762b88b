"""Kernel time of the fused table kernel against the number of frames (config-4 shape: 256 users, tile_counts=[50,100,200]):
how much of a launch is its last, partly filled wave of workgroups."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'viewport-entropy-toolkit_amd'))
import torch
from viewport_entropy_toolkit import _native, _quantiser
import bench
dev = torch.device('cuda', 0)
eng = _native.Engine(0)
U = int(sys.argv[1]) if len(sys.argv) > 1 else 256
tcs = [50, 100, 200]
plan = _native.Plan(eng, [_quantiser.lattice_xyz(t) for t in tcs], 120.0, 2.0, True, 100, 200)
mu_h, mv_h = bench.synth_video(U, 16384, 1234, 0)
mu = torch.from_numpy(mu_h).to(dev); mv = torch.from_numpy(mv_h).to(dev)
ent = torch.empty(16384, dtype=torch.float64, device=dev); idx = torch.empty((16384, U), dtype=torch.int32, device=dev)
st = torch.zeros(2, dtype=torch.int32, device=dev)
s = torch.cuda.Stream(device=dev)
for T in (3584, 4096, 7168, 8192, 9000, 10000, 10752, 12288, 14336, 16384):
    def step():
        plan.spatial_device(mu.data_ptr(), mv.data_ptr(), U, T, ent.data_ptr(), d_assign=idx.data_ptr(), d_status=st.data_ptr(), stream=s.cuda_stream)
    for _ in range(3): step()
    torch.cuda.synchronize()
    eng.profile_enable(True); eng.profile_reset()
    for _ in range(20): step()
    torch.cuda.synchronize()
    ms, n = eng.profile_get('k_spatial'); eng.profile_enable(False)
    print(f"T={T:6d} kernel {ms / n * 1e3:8.1f} us  {ms / n * 1e6 / T:7.2f} ns per frame", flush=True)
