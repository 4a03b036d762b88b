"""Kernel time of the three weighted formulations at BASELINE config-3 shape (1024 users x 30000 frames, 501 tiles)
for EntropyConfigs that make the engine choose each of them (hipEvent time of k_spatial, inputs resident in HBM)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "viewport-entropy-toolkit_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch
from bench import synth_video
from viewport_entropy_toolkit import _native, _quantiser

U, T = 1024, 30000
mu_h, mv_h = synth_video(U, T, 1234, 0)
dev = torch.device("cuda", 0)
mu, mv = torch.from_numpy(mu_h).to(dev), torch.from_numpy(mv_h).to(dev)
ent = torch.empty(T, dtype=torch.float64, device=dev)
idx = torch.empty((T, U), dtype=torch.int32, device=dev)
eng = _native.Engine(0)
for fov, power, policy in ((120.0, 2.0, 0), (120.0, 2.0, -1), (120.0, 20.0, 0), (30.0, 2.0, 0), (10.0, 2.0, 0), (120.0, 50.0, 0)):
    plan = _native.Plan(eng, [_quantiser.lattice_xyz(500)], fov, power, True, 100, 200)
    plan.set_table_policy(policy)
    for _ in range(2):
        plan.spatial_device(mu.data_ptr(), mv.data_ptr(), U, T, ent.data_ptr(), d_assign=idx.data_ptr())
    eng.synchronize()
    eng.profile_enable(True); eng.profile_reset()
    for _ in range(3):
        plan.spatial_device(mu.data_ptr(), mv.data_ptr(), U, T, ent.data_ptr(), d_assign=idx.data_ptr())
    eng.synchronize()
    ms, n = eng.profile_get("k_spatial")
    eng.profile_enable(False)
    tab, sweep = plan.error_bounds(0)
    print(f"fov={fov} power={power} policy={policy:+d}: {plan.last_formulation(0):8s} {ms / n:8.3f} ms  {U * T / (ms / n * 1e-3):.3g} samples/s  "
          f"bounds table={tab:.2e} sweep={sweep:.2e}", flush=True)
    plan.close()
