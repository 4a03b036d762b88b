"""GPU: the 1e-6 RELATIVE entropy contract where it is hardest — frames of one or two users, whose
entropy can be arbitrarily small — over the whole discrete sample domain (all 20 301 pixel directions),
for EntropyConfigs whose weights span many orders of magnitude (narrow FoV, large power factor);
degenerate lattices of 1-3 tiles; a BASELINE-config-4-shaped video.  Reference contract:
utilities/entropy_utils.py:124-137 (weights), 194-209 (entropy and normaliser)."""
import numpy as np
import pytest

from oracle import vet_oracle as vo
from tests._tol import W_RTOL, w_atol

pytestmark = pytest.mark.gpu

W, H = 100, 200


@pytest.fixture(scope="module")
def native():
    from viewport_entropy_toolkit import _native
    return _native


@pytest.fixture(scope="module")
def engine(native):
    return native.Engine.default()


def all_directions():
    """One sample per pixel direction: mu/mv that truncate to every (px, py) of the 101 x 201 grid."""
    px, py = np.meshgrid(np.arange(W + 1), np.arange(H + 1))
    px, py = px.ravel(), py.ravel()
    mu = np.where(px == W, 1.0, (px + 0.5) / W)
    mv = np.where(py == H, 1.0, (py + 0.5) / H)
    return mu, mv


# The contract is relative (1e-6).  The absolute term only covers the FP64 rounding of the reference's own
# evaluation: with p1 = 1 - O(1e-12) the term -p1*log2(p1) of entropy_utils.py:196-198 moves by 1.6e-16 per ulp
# of p1, and numpy's and the device's log2 / summation order differ by such ulps (oracle vs the live
# reference differ by as much: tests/golden was pinned at <= 1e-15).  It is 9 orders below an entropy of 1e-6.
ATOL = 1e-15
EXTREME = [([50], 30.0, 2.0), ([50], 120.0, 20.0), ([500], 5.0, 3.0), ([500], 10.0, 2.0), ([50], 120.0, 50.0)]
# power factors whose weights underflow: the reference's NaN frames (entropy_utils.py:131-135, 195-198; golden G12)
UNDERFLOW = [([500], 120.0, 100.0), ([50], 120.0, 200.0), ([500], 60.0, 150.0)]
_ORACLE = {}


def contract_cases():
    mu1, mv1 = all_directions()
    rng = np.random.default_rng(20301)
    perm = rng.permutation(len(mu1))
    near = (np.arange(len(mu1)) + 1) % len(mu1)                # neighbouring pixel: overlapping cones
    return {
        "one user": (mu1[:, None], mv1[:, None]),
        "two users, random pair": (np.stack([mu1, mu1[perm]], 1), np.stack([mv1, mv1[perm]], 1)),
        "two users, neighbours": (np.stack([mu1, mu1[near]], 1), np.stack([mv1, mv1[near]], 1)),
    }


def oracle_for(tcs, fov, power, name, mu, mv):
    key = (tuple(tcs), fov, power, name)
    if key not in _ORACLE:
        _ORACLE[key] = vo.spatial_series(mu, mv, W, H, tcs, fov_angle=fov, power_factor=power)[:2]
    return _ORACLE[key]


@pytest.mark.parametrize("tcs,fov,power", EXTREME + UNDERFLOW + [([500], 120.0, 2.0), ([20, 50], 120.0, 2.0)])
@pytest.mark.parametrize("policy", [1, -1, 0])
def test_single_and_two_user_frames_over_all_directions(native, engine, tcs, fov, power, policy):
    cases = contract_cases()
    plan = native.Plan(engine, [vo.fibonacci_lattice(tc) for tc in tcs], fov, power, True, W, H)
    plan.set_table_policy(policy)
    tab_bound, sweep_bound = plan.error_bounds(0)
    for name, (mu, mv) in cases.items():
        res = plan.spatial(mu=mu, mv=mv)
        form = plan.last_formulation(0)
        # an integer formulation only where the plan's own bound puts it inside the contract
        table_asked = policy > 0 or (policy == 0 and mu.size >= native.TABLE_SAMPLES_PER_DIRECTION * (W + 1) * (H + 1))
        want = ("table" if tab_bound <= 1e-7 else "ftable") if table_asked else ("sweep" if sweep_bound <= 1e-7 else "precise")
        assert form == want, (name, form, tab_bound, sweep_bound)
        ent, assign = oracle_for(tcs, fov, power, name, mu, mv)
        assert np.array_equal(res["assign"], assign), name
        assert np.array_equal(np.isnan(res["entropy"]), np.isnan(ent)), name
        ok = ~np.isnan(ent)
        np.testing.assert_allclose(res["entropy"][ok], ent[ok], rtol=1e-6, atol=ATOL, err_msg=f"{name} [{form}]")
    plan.close()


@pytest.mark.parametrize("tcs", [[1], [2], [3], [1, 3], [3, 50], [50, 3]])
@pytest.mark.parametrize("policy", [1, -1, 0])
@pytest.mark.parametrize("fov,power", [(120.0, 2.0), (30.0, 0.5), (360.0, 1.0)])
def test_degenerate_lattices(native, engine, tcs, policy, fov, power):
    """1-, 3- and 3-tile lattices (tile_count 1, 2, 3 -> n = 1, 3, 3): frames whose users weigh on one tile
    only, frames where nobody has a tile in the FoV, the 0 / -0.0 normaliser of a one-tile lattice; the
    nan / 0.0 pattern must be the reference's."""
    rng = np.random.default_rng(sum(tcs) * 7 + int(fov))
    U, T = 61, 48
    mu, mv = rng.random((T, U)), rng.random((T, U))
    gone = rng.random((T, U)) < 0.3
    gone[:, 0] = False
    mu[gone] = np.nan
    mv[gone] = np.nan
    mu[:8, 1:] = np.nan                     # single-user frames
    mv[:8, 1:] = np.nan
    plan = native.Plan(engine, [vo.fibonacci_lattice(tc) for tc in tcs], fov, power, True, W, H)
    plan.set_table_policy(policy)
    res = plan.spatial(mu=mu, mv=mv, want_weights=True)
    ent, assign, weights = vo.spatial_series(mu, mv, W, H, tcs, fov_angle=fov, power_factor=power, want_weights=True)
    assert np.array_equal(res["assign"], assign)
    assert np.array_equal(np.isnan(res["entropy"]), np.isnan(ent)), (res["entropy"][:10], ent[:10])
    ok = ~np.isnan(ent)
    np.testing.assert_allclose(res["entropy"][ok], ent[ok], rtol=1e-6, atol=ATOL)
    # tile_weights values: the reference's under every formulation (tests/_tol.py)
    np.testing.assert_allclose(res["weights"], weights, rtol=W_RTOL, atol=w_atol(U, power))
    plan.close()


@pytest.mark.parametrize("policy", [1, -1, 0])
def test_config4_shape(native, engine, policy):
    """BASELINE config 4, one GPU's share: 256 users x 10 000 frames, tile_counts=[50,100,200]: run-to-run
    bit equality, invariance under user permutation and frame split (integer histograms), nearest == LUT,
    sampled frames against the oracle."""
    U, T, tcs = 256, 10000, [50, 100, 200]
    rng = np.random.default_rng(44)
    mu = np.mod(0.5 + np.cumsum(rng.normal(0, 0.01, (T, U)), axis=0), 1.0)
    mv = np.clip(0.5 + np.cumsum(rng.normal(0, 0.005, (T, U)), axis=0), 0.0, 1.0)
    plan = native.Plan(engine, [vo.fibonacci_lattice(tc) for tc in tcs], 120.0, 2.0, True, W, H)
    plan.set_table_policy(policy)
    a = plan.spatial(mu=mu, mv=mv, want_assign=True, want_weights=True)
    assert plan.last_formulation(0) == ("sweep" if policy < 0 else "table")
    b = plan.spatial(mu=mu, mv=mv, want_assign=False)
    assert np.array_equal(a["entropy"], b["entropy"])
    assert np.all(np.isfinite(a["entropy"])) and a["entropy"].min() > 0 and a["entropy"].max() <= 1.0
    perm = rng.permutation(U)
    c = plan.spatial(mu=mu[:, perm], mv=mv[:, perm], want_assign=False)
    assert np.array_equal(c["entropy"], a["entropy"])
    d = plan.spatial(mu=mu[3333:7001], mv=mv[3333:7001], want_assign=False)
    if policy != 0:                      # under policy 0 the shorter call is still a table call (2 x 20 301 samples)
        assert np.array_equal(d["entropy"], a["entropy"][3333:7001])
    else:
        np.testing.assert_allclose(d["entropy"], a["entropy"][3333:7001], rtol=1e-8)
    near = plan.read_nearest(0).reshape(H + 1, W + 1)
    assert np.array_equal(a["assign"], near[(mv * H).astype(int), (mu * W).astype(int)])
    frames = np.concatenate([[0, T - 1], rng.integers(0, T, 30)])
    ent, assign, weights = vo.spatial_series(mu[frames], mv[frames], W, H, tcs, want_weights=True)
    assert np.array_equal(a["assign"][frames], assign)
    np.testing.assert_allclose(a["entropy"][frames], ent, rtol=1e-8)
    np.testing.assert_allclose(a["weights"][frames], weights, rtol=W_RTOL, atol=w_atol(U))
    plan.close()


def test_error_bounds_of_the_baseline_plans(native, engine):
    """The BASELINE configurations (fov 120, power 2, lattices of 21 ... 1001 tiles) are provably inside the
    contract with integer histograms; narrow cones and large powers are not."""
    plan = native.Plan(engine, [vo.fibonacci_lattice(tc) for tc in (20, 50, 100, 200, 250, 500, 1000)], 120.0, 2.0,
                       True, W, H)
    for k in range(7):
        tab, sweep = plan.error_bounds(k)
        assert 0 < tab <= 1e-7 and 0 < sweep <= 1e-7, (k, tab, sweep)
    plan.close()
    for tcs, fov, power in EXTREME:
        plan = native.Plan(engine, [vo.fibonacci_lattice(tc) for tc in tcs], fov, power, True, W, H)
        tab, sweep = plan.error_bounds(0)
        if (tcs, fov, power) == ([500], 5.0, 3.0):
            assert tab == np.inf and sweep == np.inf          # every row has at most one tile in its FoV
        else:
            assert tab > 1e-7
        plan.set_table_policy(1)                              # ... so the table such a plan builds holds FP32 weights
        plan.spatial(mu=np.full((2, 3), 0.4), mv=np.full((2, 3), 0.6))
        assert plan.last_formulation(0) == "ftable"
        plan.close()


@pytest.mark.parametrize("tcs,fov,power", [([500], 10.0, 2.0), ([50], 120.0, 20.0), ([50, 100], 120.0, 30.0), ([500], 120.0, 150.0)])
def test_fp_table_is_bit_reproducible(native, engine, tcs, fov, power):
    """`ftable` (FP32 table weights, FP64 histograms): every frame's row list is sorted and every wave adds its rows into
    a histogram of its own in program order, so the floats do not depend on scheduling — bit-identical run to run, under
    any permutation of the users, under a split of the frame axis (= the GPU count).  Plans with marker entries hand
    their undecided frames to the precise sweep, which then sums the users in ascending direction order: the same holds."""
    U, T = 256, 1500
    rng = np.random.default_rng(int(fov * power))
    mu = np.mod(0.5 + np.cumsum(rng.normal(0, 0.01, (T, U)), axis=0), 1.0)
    mv = np.clip(0.5 + np.cumsum(rng.normal(0, 0.005, (T, U)), axis=0), 0.0, 1.0)
    plan = native.Plan(engine, [vo.fibonacci_lattice(tc) for tc in tcs], fov, power, True, W, H)
    plan.set_table_policy(1)
    a = plan.spatial(mu=mu, mv=mv, want_weights=True)
    assert plan.last_formulation(0) == "ftable"
    for _ in range(3):
        b = plan.spatial(mu=mu, mv=mv, want_weights=True)
        assert np.array_equal(a["entropy"], b["entropy"], equal_nan=True)
        assert np.array_equal(a["weights"], b["weights"])
    d = plan.spatial(mu=mu[377:1201], mv=mv[377:1201], want_assign=False)
    assert np.array_equal(d["entropy"], a["entropy"][377:1201], equal_nan=True)
    perm = rng.permutation(U)
    c = plan.spatial(mu=mu[:, perm], mv=mv[:, perm], want_assign=False)
    assert np.array_equal(c["entropy"], a["entropy"], equal_nan=True)
    few = plan.spatial(mu=mu[:40, :100], mv=mv[:40, :100])          # fewer than 128 users: no set of distinct rows
    few2 = plan.spatial(mu=mu[:40, :100][:, ::-1].copy(), mv=mv[:40, :100][:, ::-1].copy())
    assert np.array_equal(few["entropy"], few2["entropy"], equal_nan=True)
    frames = np.concatenate([[0, T - 1], rng.integers(0, T, 12)])
    ent, _, _ = vo.spatial_series(mu[frames], mv[frames], W, H, tcs, fov_angle=fov, power_factor=power)
    assert np.array_equal(np.isnan(a["entropy"][frames]), np.isnan(ent))
    ok = ~np.isnan(ent)
    np.testing.assert_allclose(a["entropy"][frames][ok], ent[ok], rtol=1e-6, atol=ATOL)
    plan.close()


@pytest.mark.parametrize("tcs,fov,power", [([50], 120.0, 50.0), ([50], 120.0, 20.0)])
@pytest.mark.parametrize("policy", [1, -1])
def test_tiny_entropies_against_extended_precision(native, engine, tcs, fov, power, policy):
    """Where the 1e-6 relative contract meets the FP64 rounding floor of the reference's own -p*log2(p) (single-user
    frames whose entropy is 1e-9 ... 4e-12: the ATOL term of the tests above), the comparison is made against the
    entropy evaluated in extended precision (numpy longdouble) from the same FP64 tile weights: the engine must be
    within 1e-6 relative of it, or at most twice as far from it as the FP64 reference path itself gets."""
    if np.finfo(np.longdouble).eps > 1e-18:
        pytest.skip("no extended precision on this host")
    L = vo.fibonacci_lattice(tcs[0])
    flat = vo.direction_grid(W, H).reshape(-1, 3)
    rows = vo.tile_weight_rows(flat, L, fov, power)                      # FP64 weights, as the reference computes them
    mu, mv = all_directions()
    ld = rows.astype(np.longdouble)
    tot = ld.sum(axis=1, keepdims=True)
    with np.errstate(all="ignore"):
        p_ld = ld / tot
        h_ld = -(np.where(ld > 0, p_ld * np.log2(np.where(ld > 0, p_ld, 1)), 0)).sum(axis=1) / np.log2(np.longdouble(len(L)))
    ref64, _ = oracle_for(tcs, fov, power, "one user", mu[:, None], mv[:, None])
    plan = native.Plan(engine, [L], fov, power, True, W, H)
    plan.set_table_policy(policy)
    got = plan.spatial(mu=mu[:, None], mv=mv[:, None], want_assign=False)["entropy"]
    plan.close()
    ok = np.isfinite(ref64) & (h_ld > 0)
    err_eng = np.abs(got.astype(np.longdouble) - h_ld)[ok]
    err_ref = np.abs(ref64.astype(np.longdouble) - h_ld)[ok]
    rel_ok = err_eng <= 1e-6 * h_ld[ok]
    tiny = ~rel_ok
    # the frames outside 1e-6 relative are exactly the ones where FP64 itself cannot do better
    floor = float(err_ref.max())
    assert floor < 1e-15
    assert np.all(err_eng[tiny] <= 2 * floor + 1e-22), (int(tiny.sum()), float(err_eng[tiny].max()), floor)
    assert np.all(h_ld[ok][tiny] < 1e-8)
