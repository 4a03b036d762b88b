"""GPU: the reference's NaN frames.  A power factor large enough makes ((max - d) / max) ** power underflow to
exactly 0.0 for tiles that ARE in a user's FoV; the reference keeps them as keys of its per-frame dict
(utilities/entropy_utils.py:131-135) and 0 * log2(0) makes the frame's entropy NaN (:195-198).  Golden G12 was
produced by the live reference (oracle/gen_golden.py); every formulation the engine can choose for such plans
(`precise`, `ftable` + the in-call resolver) must reproduce the NaN pattern, the finite values and the key set
of ``tile_weights`` (0.0-valued keys included)."""
import numpy as np
import pytest

from oracle import vet_oracle as vo

pytestmark = pytest.mark.gpu

W, H = 100, 200
CONFIGS = [(tc, fov, power) for tc in (50, 500) for fov in (120, 60) for power in (50, 80, 100, 150, 200)]


@pytest.fixture(scope="module")
def native():
    from viewport_entropy_toolkit import _native
    return _native


@pytest.fixture(scope="module")
def engine(native):
    return native.Engine.default()


def samples(px, py):
    present = px >= 0
    mu = np.where(present, np.where(px == W, 1.0, (px + 0.5) / W), np.nan)
    mv = np.where(present, np.where(py == H, 1.0, (py + 0.5) / H), np.nan)
    return mu, mv


def keys_of(weights):
    return (weights != 0) | np.signbit(weights)


@pytest.mark.parametrize("raw", [pytest.param(False, id="weights-pass"), pytest.param(True, id="own-histogram")])
@pytest.mark.parametrize("policy", [1, -1, 0])
@pytest.mark.parametrize("tc,fov,power", CONFIGS)
def test_g12_nan_pattern_and_keys(native, engine, golden_dir, tc, fov, power, policy, raw):
    g = np.load(golden_dir / "g12_underflow.npz")
    tag = f"tc{tc}_fov{fov}_p{power}"
    mu, mv = samples(g["px"], g["py"])
    plan = native.Plan(engine, [vo.fibonacci_lattice(tc)], float(fov), float(power), True, W, H)
    plan.set_table_policy(policy)
    plan.set_raw_weights(raw)
    tab_bound, sweep_bound = plan.error_bounds(0)
    res = plan.spatial(mu=mu, mv=mv, want_weights=True)
    form = plan.last_formulation(0)
    want = ("table" if tab_bound <= 1e-7 else "ftable") if policy > 0 else ("sweep" if sweep_bound <= 1e-7 else "precise")
    assert form == want, (form, tab_bound, sweep_bound)
    ref = g[f"{tag}__entropy"]
    if np.isnan(ref).any() or (g[f"{tag}__keys"] & (g[f"{tag}__hist"] < 1e-300)).any():
        assert form in ("ftable", "precise") and tab_bound == np.inf      # weights underflow: FP64 formulations only
    assert np.array_equal(np.isnan(res["entropy"]), np.isnan(ref)), (form, np.flatnonzero(np.isnan(res["entropy"]) != np.isnan(ref)))
    ok = ~np.isnan(ref)
    np.testing.assert_allclose(res["entropy"][ok], ref[ok], rtol=1e-6, atol=1e-15)
    if not raw:
        # the weights output proper: the reference's keys and values (subnormal weights included) under every formulation
        assert np.array_equal(keys_of(res["weights"]), g[f"{tag}__keys"]), form
        np.testing.assert_allclose(np.abs(res["weights"]), g[f"{tag}__hist"], rtol=1e-9, atol=0)
    elif form in ("ftable", "precise"):                # the FP64 formulations keep the reference's key set exactly
        assert np.array_equal(keys_of(res["weights"]), g[f"{tag}__keys"]), form
        np.testing.assert_allclose(np.abs(res["weights"]), g[f"{tag}__hist"], rtol=2e-7 if form == "ftable" else 1e-9,
                                   atol=8 * 2.0 ** -149 if form == "ftable" else 0)     # FP32 entries below 2^-126 of their row's scale are denormal
    else:
        np.testing.assert_allclose(res["weights"], g[f"{tag}__hist"], rtol=1e-9, atol=2.0 ** -33 * 8)
    plan.close()


@pytest.mark.parametrize("tcs,fov,power,U", [([500], 120.0, 150.0, 160), ([50, 500], 120.0, 100.0, 160), ([50, 500], 60.0, 200.0, 48),
                                             ([500, 50], 120.0, 80.0, 200)])
@pytest.mark.parametrize("raw", [pytest.param(False, id="weights-pass"), pytest.param(True, id="own-histogram")])
@pytest.mark.parametrize("policy", [1, -1])
def test_resolver_on_crowded_frames(native, engine, tcs, fov, power, U, policy, raw):
    """Frames of many users (the table kernel's set of distinct rows, fused lattices): the frames the marker
    entries cannot decide go through the precise sweep inside the call and come back as the oracle has them."""
    rng = np.random.default_rng(int(power) + U)
    T = 96
    px = rng.integers(0, W + 1, (T, U))
    py = np.clip(np.rint(100 + 40 * rng.standard_normal((T, U))), 0, H).astype(np.int64)
    px[: T // 2] = (px[: T // 2, :1] + rng.integers(-10, 11, (T // 2, U))) % (W + 1)      # clustered half
    py[: T // 2] = np.clip(py[: T // 2, :1] + rng.integers(-10, 11, (T // 2, U)), 0, H)
    gone = rng.random((T, U)) < 0.2
    gone[:, 0] = False
    px[gone] = -1
    mu, mv = samples(px, py)
    plan = native.Plan(engine, [vo.fibonacci_lattice(tc) for tc in tcs], fov, power, True, W, H)
    plan.set_table_policy(policy)
    plan.set_raw_weights(raw)
    a = plan.spatial(mu=mu, mv=mv, want_weights=True)
    assert plan.last_formulation(0) == ("ftable" if policy > 0 else "precise")
    b = plan.spatial(mu=mu, mv=mv, want_weights=False, want_assign=False)
    assert np.array_equal(np.isnan(a["entropy"]), np.isnan(b["entropy"]))
    ent, assign, weights = vo.spatial_series(mu, mv, W, H, tcs, fov_angle=fov, power_factor=power, want_weights=True)
    assert np.array_equal(a["assign"], assign)
    assert np.array_equal(np.isnan(a["entropy"]), np.isnan(ent)), np.flatnonzero(np.isnan(a["entropy"]) != np.isnan(ent))
    ok = ~np.isnan(ent)
    np.testing.assert_allclose(a["entropy"][ok], ent[ok], rtol=1e-6, atol=1e-15)
    assert np.array_equal(keys_of(a["weights"]), keys_of(weights))
    fp32 = raw and policy > 0                      # the FP table's own histogram: FP32 entries; otherwise the reference's values
    np.testing.assert_allclose(np.abs(a["weights"]), np.abs(weights), rtol=2e-7 if fp32 else 1e-9, atol=U * 2.0 ** -149 if fp32 else 0)
    plan.close()


def test_operator_level_nan_and_zero_valued_keys(golden_dir):
    """compute_spatial_entropy / calculate_tile_weights (the reference's function-level boundary) on G12 frames."""
    from viewport_entropy_toolkit import Vector
    from viewport_entropy_toolkit.utilities import (EntropyConfig, calculate_tile_weights, compute_spatial_entropy,
                                                    generate_fibonacci_lattice)
    g = np.load(golden_dir / "g12_underflow.npz")
    grid = np.load(golden_dir / "g2_quantiser.npz")["vec_100x200"]
    L = generate_fibonacci_lattice(500)
    cfg = EntropyConfig(fov_angle=120.0, power_factor=150.0)
    tag = "tc500_fov120_p150"
    for f in (0, 3, 12, 25, 38):
        users = {f"u{u}": Vector(*map(float, grid[g["py"][f, u], g["px"][f, u]])) for u in range(8) if g["px"][f, u] >= 0}
        e, tw, _ = compute_spatial_entropy(users, L, cfg)
        ref = g[f"{tag}__entropy"][f]
        assert np.isnan(e) == np.isnan(ref)
        if not np.isnan(ref):
            assert abs(e - ref) <= 1e-6 * abs(ref) + 1e-15
        keys = np.zeros(len(L), dtype=bool)
        idx = {v: i for i, v in enumerate(L)}
        for v in tw:
            keys[idx[v]] = True
        assert np.array_equal(keys, g[f"{tag}__keys"][f])
        assert all(w >= 0 and not np.signbit(w) for w in tw.values())
    one = calculate_tile_weights(Vector(*map(float, grid[g["py"][25, 0], g["px"][25, 0]])), L, cfg)
    assert len(one) == int(g[f"{tag}__keys"][25].sum())
