"""Time of the weights-only pass (k_weights_gather over the exact FP64 rows of lattice 0; vet_host.hpp: WeightsCore) on a
config-3-shaped video: the eager d_weights output for T frames, and the fetch of weight-row blocks of a resident result.
usage: python tools/weights_pass_timing.py"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'viewport-entropy-toolkit_amd'))
import numpy as np
import torch
from viewport_entropy_toolkit import _native, _quantiser
import bench

dev = torch.device('cuda', 0)
eng = _native.Engine(0)
U, n = 1024, 501
mu_h, mv_h = bench.synth_video(U, 30000, 1234, 0)
plan = _native.Plan(eng, [_quantiser.lattice_xyz(500)], 120.0, 2.0, True, 100, 200)
mu = torch.from_numpy(mu_h).to(dev); mv = torch.from_numpy(mv_h).to(dev)
ent = torch.empty(30000, dtype=torch.float64, device=dev)
wts = torch.empty((30000, n), dtype=torch.float64, device=dev)
st = torch.zeros(2, dtype=torch.int32, device=dev)
stream = torch.cuda.Stream(device=dev)
for T in (1, 256, 4096, 30000):
    def step():
        plan.spatial_device(mu.data_ptr(), mv.data_ptr(), U, T, ent.data_ptr(), d_weights=wts.data_ptr(), d_status=st.data_ptr(),
                            stream=stream.cuda_stream)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    eng.profile_enable(True); eng.profile_reset()
    for _ in range(5):
        step()
    torch.cuda.synchronize()
    ms_w, n_w = eng.profile_get('k_weights'); ms_s, n_s = eng.profile_get('k_spatial')
    eng.profile_enable(False)
    print(f"eager T={T:6d}: k_weights {ms_w / max(n_w, 1):8.3f} ms ({n_w} launches)   k_spatial {ms_s / max(n_s, 1):8.3f} ms   formulation {plan.last_formulation(0)}", flush=True)
r = plan.spatial_resident(mu=mu_h, mv=mv_h)
res = r["result"]
for nrows in (1, 256, 256, 1024, 4096, 4096, 256):
    t0 = time.perf_counter(); a = res.rows(1, 2000, nrows); dt = time.perf_counter() - t0
    t0 = time.perf_counter(); b = res.rows(0, 2000, nrows); dt0 = time.perf_counter() - t0
    print(f"fetch rows={nrows:5d}: weights {dt * 1e3:8.3f} ms ({a.nbytes >> 10} KiB)   assignments {dt0 * 1e3:8.3f} ms ({b.nbytes >> 10} KiB)", flush=True)
