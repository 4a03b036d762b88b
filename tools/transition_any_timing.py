"""Transition mode beyond 4 096 users (k_transition_big: bucket hash in LDS, the row cut into ranges of source tiles; round 3:
k_transition_any with the hash in global scratch) at 9 000 and 20 000 users, 200 tiles: kernel time and algorithmic GB/s
(24 B per sample + 8 B per row).
usage: python tools/transition_any_timing.py    (under rocprofv3 --kernel-trace --stats for the committed summary)"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'viewport-entropy-toolkit_amd'))
import numpy as np
import torch
from viewport_entropy_toolkit import _native, _quantiser
import bench

dev = torch.device('cuda', 0)
eng = _native.Engine(0)
plan = _native.Plan(eng, [_quantiser.lattice_xyz(200)], 120.0, 2.0, True, 100, 200)
# (users, frames, first frame of the slice): the random-walk audience starts in ONE direction and spreads with time, so the
# LATE rows of a long video have more distinct (source, destination) pairs — more ranges (passes) per row in
# k_transition_big — than its first rows: the same 256 rows cost more at frames 768.. than at frames 0..
for U, T, first in ((4096, 513, 0), (9000, 513, 0), (20000, 257, 0), (9000, 2049, 0), (20000, 1025, 0), (20000, 257, 768),
                    (20000, 257, 3840), (9000, 513, 1536)):
    mu_h, mv_h = bench.synth_video(U, first + T, 1234, 0)
    mu_h, mv_h = np.ascontiguousarray(mu_h[first:]), np.ascontiguousarray(mv_h[first:])
    mu = torch.from_numpy(mu_h).to(dev); mv = torch.from_numpy(mv_h).to(dev)
    ent = torch.empty(T, dtype=torch.float64, device=dev); idx = torch.empty((T, U, 2), dtype=torch.int32, device=dev)
    st = torch.zeros(2, dtype=torch.int32, device=dev)
    run_stream = torch.cuda.Stream(device=dev)

    def step():
        plan.transition_device(mu.data_ptr(), mv.data_ptr(), U, T, ent.data_ptr(), d_pairs=idx.data_ptr(),
                               d_status=st.data_ptr(), stream=run_stream.cuda_stream)
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    eng.profile_enable(True); eng.profile_reset()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    ms, n = eng.profile_get('k_transition')
    eng.profile_enable(False)
    alg = 24.0 * U * (T - 1) + 8.0 * (T - 1)
    print(f"U={U:6d} rows={T - 1:4d} from frame {first:4d} kernel {ms / n:8.3f} ms  {U * (T - 1) / (ms / n * 1e-3):.3e} pair-samples/s  "
          f"{alg / (ms / n * 1e-3) / 1e9:7.1f} GB/s algorithmic ({alg / (ms / n * 1e-3) / 8e12:.3f} of 8 TB/s)"
          f"  kernel={'k_transition_run (registers + LDS)' if U <= 4096 else 'k_transition_big (LDS hash, source-tile ranges)'}", flush=True)
