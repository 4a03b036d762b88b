"""Deviation of the engine from the oracle on the hardest frames of the 1e-6 relative contract: one and two users over all
20 301 pixel directions, for benign and extreme EntropyConfigs, per table policy (tests/test_hip_contract.py asserts the
same data at rtol 1e-6).  Prints one line per (config, policy): formulation, proven bounds, worst relative deviation."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "viewport-entropy-toolkit_amd"), str(ROOT / "tests")):
    sys.path.insert(0, p)
import numpy as np
from oracle import vet_oracle as vo
from viewport_entropy_toolkit import _native
import test_hip_contract as t

eng = _native.Engine(0)
cases = t.contract_cases()
for tcs, fov, power in t.EXTREME + [([500], 120.0, 2.0), ([20, 50], 120.0, 2.0), ([1000], 120.0, 2.0)]:
    for policy in (1, -1):
        plan = _native.Plan(eng, [vo.fibonacci_lattice(tc) for tc in tcs], fov, power, True, 100, 200)
        plan.set_table_policy(policy)
        tab, sweep = plan.error_bounds(0)
        worst, worst_big, forms = 0.0, 0.0, set()
        for name, (mu, mv) in cases.items():
            res = plan.spatial(mu=mu, mv=mv)
            forms.add(plan.last_formulation(0))
            ent, _ = t.oracle_for(tcs, fov, power, name, mu, mv)
            ok = ~np.isnan(ent) & (ent != 0)
            rel = np.abs(res["entropy"][ok] - ent[ok]) / np.abs(ent[ok])
            worst = max(worst, rel.max() if rel.size else 0.0)
            big = ok & (ent > 1e-9)
            relb = np.abs(res["entropy"][big] - ent[big]) / np.abs(ent[big])
            worst_big = max(worst_big, relb.max() if relb.size else 0.0)
        print(f"tile_counts={tcs} fov={fov} power={power} policy={policy:+d}: {sorted(forms)} bounds table={tab:.2e} sweep={sweep:.2e} "
              f"worst rel dev {worst:.2e} (entropies > 1e-9: {worst_big:.2e})", flush=True)
        plan.close()
