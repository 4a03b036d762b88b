"""GPU: odd shapes and sizes against the oracle — user counts that exercise the LDS chunking,
odd / tiny user counts, 1-frame videos, the reference's five default lattices in one call, tiny
and large pixel grids, large transition frames, absent users everywhere."""
import numpy as np
import pytest

from oracle import vet_oracle as vo
from tests._tol import W_RTOL, w_atol

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def native():
    from viewport_entropy_toolkit import _native
    return _native


@pytest.fixture(scope="module")
def engine(native):
    return native.Engine.default()


def video(U, T, seed, p_absent=0.1):
    rng = np.random.default_rng(seed)
    mu = np.mod(0.5 + np.cumsum(rng.normal(0, 0.02, (T, U)), axis=0) + rng.random((1, U)), 1.0)
    mv = np.clip(0.5 + np.cumsum(rng.normal(0, 0.01, (T, U)), axis=0) + rng.normal(0, 0.2, (1, U)), 0.0, 1.0)
    gone = rng.random((T, U)) < p_absent
    gone[np.arange(T), rng.integers(0, U, T)] = False
    mu[gone] = np.nan
    mv[gone] = np.nan
    return mu, mv


def plan_for(native, engine, tcs, W=100, H=200, weighted=True, policy=0, **kw):
    plan = native.Plan(engine, [vo.fibonacci_lattice(tc) for tc in tcs], kw.get("fov", 120.0), kw.get("power", 2.0),
                       weighted, W, H)
    plan.set_table_policy(policy)
    return plan


@pytest.mark.parametrize("U,T,tcs,weighted,policy", [
    (3000, 7, [50], True, 1),          # more users than one LDS id chunk (table formulation)
    (2500, 5, [100], True, -1),        # more users than one LDS direction chunk (sweep formulation)
    (3000, 6, [50], False, 0),         # unweighted, persistent LDS-LUT kernel, many users
    (1001, 9, [50], False, 0),         # odd user count: generic unweighted kernel
    (7, 40, [20, 50], True, 1),
    (7, 40, [20, 50], True, -1),
    (1, 1, [50], True, 0),
    (2, 3, [1], True, 0),              # one-tile lattice: nan everywhere
    (33, 17, [2], False, 0),
    (40, 50, [20, 50, 100, 250, 1000], True, 1),     # the reference's default tile_counts, fused
    (40, 50, [20, 50, 100, 250, 1000], True, -1),
    (40, 50, [20, 50, 100, 250, 1000], False, 0),
    (64, 30, [1000], True, 1),
    (600, 4, [2000], True, -1),        # n = 2001: 16 tile groups of 128, several per wave
])
def test_spatial_shapes(native, engine, U, T, tcs, weighted, policy):
    mu, mv = video(U, T, seed=U * 31 + T)
    plan = plan_for(native, engine, tcs, weighted=weighted, policy=policy)
    res = plan.spatial(mu=mu, mv=mv, want_weights=True)
    ent, assign, weights = vo.spatial_series(mu, mv, 100, 200, tcs, use_weight_distribution=weighted,
                                             want_weights=True)
    assert np.array_equal(res["assign"], assign)
    assert np.array_equal(res["present"], (~np.isnan(mu)).sum(1))
    np.testing.assert_allclose(res["entropy"], ent, rtol=1e-8, equal_nan=True)
    np.testing.assert_allclose(res["weights"], weights, rtol=W_RTOL, atol=w_atol(U))
    plan.close()


@pytest.mark.parametrize("W,H,tc", [(6, 4, 20), (640, 480, 50), (3840, 1920, 20)])
@pytest.mark.parametrize("weighted", [True, False])
def test_other_pixel_grids(native, engine, W, H, tc, weighted):
    mu, mv = video(48, 64, seed=W + H)
    plan = plan_for(native, engine, [tc], W, H, weighted=weighted, policy=-1 if W > 1000 else 0)
    res = plan.spatial(mu=mu, mv=mv)
    ent, assign, _ = vo.spatial_series(mu, mv, W, H, [tc], use_weight_distribution=weighted)
    assert np.array_equal(res["assign"], assign)
    np.testing.assert_allclose(res["entropy"], ent, rtol=1e-8, equal_nan=True)
    tr = plan.transition(mu=mu, mv=mv, check=False)
    e2, pairs = vo.transition_series(mu, mv, W, H, [tc]) if (tr["common"] > 0).all() else (None, None)
    if e2 is not None:
        assert np.array_equal(tr["pairs"], pairs)
        np.testing.assert_allclose(tr["entropy"], e2, rtol=1e-9, equal_nan=True)
    plan.close()


@pytest.mark.parametrize("U,T,tcs", [(1500, 6, [50]), (4096, 3, [20]), (3, 30, [20, 50, 100]), (2, 2, [1000]),
                                     (9000, 3, [20]), (20000, 2, [200]), (5000, 4, [50, 20]),
                                     # lanes without a user in the run kernel (1, 2 and 4 users per lane), an exact fit
                                     (300, 9, [200]), (129, 5, [50]), (640, 4, [100]), (512, 7, [200]), (64, 40, [20])])
def test_transition_shapes(native, engine, U, T, tcs):
    mu, mv = video(U, T, seed=U + 7 * T, p_absent=0.05)
    mu[:, 0] = np.where(np.isnan(mu[:, 0]), 0.5, mu[:, 0])      # user 0 always present: no empty rows
    mv[:, 0] = np.where(np.isnan(mv[:, 0]), 0.5, mv[:, 0])
    plan = plan_for(native, engine, tcs)
    res = plan.transition(mu=mu, mv=mv, want_srccount=True)
    ent, pairs = vo.transition_series(mu, mv, 100, 200, tcs, closed_form=False)
    assert np.array_equal(res["pairs"], pairs)
    assert np.array_equal(res["common"], (~np.isnan(mu[1:]) & ~np.isnan(mu[:-1])).sum(1))
    np.testing.assert_allclose(res["entropy"], ent, rtol=1e-9, equal_nan=True)
    plan.close()


def test_transition_global_hash_variant_on_small_frames(native, engine, monkeypatch):
    """The fallback of k_transition for lattices of thousands of tiles (bucket hash in global scratch, persistent
    workgroups) forced onto ordinary shapes: same results as the LDS variant, rows looped per workgroup.  The tuning
    environment is read when an engine is created: the forced variant runs on an engine of its own."""
    mu, mv = video(300, 700, seed=77, p_absent=0.1)
    mu[:, 0] = np.where(np.isnan(mu[:, 0]), 0.5, mu[:, 0])
    mv[:, 0] = np.where(np.isnan(mv[:, 0]), 0.5, mv[:, 0])
    plan = plan_for(native, engine, [50, 200])
    a = plan.transition(mu=mu, mv=mv, want_srccount=True)
    plan.close()
    monkeypatch.setenv("VET_T_GLOBAL", "1")
    eng2 = native.Engine(0)
    monkeypatch.delenv("VET_T_GLOBAL")
    plan = plan_for(native, eng2, [50, 200])
    b = plan.transition(mu=mu, mv=mv, want_srccount=True)
    plan.close()
    eng2.close()
    for k in ("pairs", "srccount", "common"):
        assert np.array_equal(a[k], b[k]), k
    np.testing.assert_allclose(b["entropy"], a["entropy"], rtol=1e-13, equal_nan=True)
    ent, pairs = vo.transition_series(mu, mv, 100, 200, [50, 200])
    assert np.array_equal(a["pairs"], pairs)
    np.testing.assert_allclose(a["entropy"], ent, rtol=1e-9, equal_nan=True)


@pytest.mark.parametrize("U,T,tc,kind", [(9000, 4, 200, "walk"), (20000, 3, 200, "uniform"), (4097, 5, 20, "walk"),
                                         (6000, 3, 1000, "uniform"), (5000, 4, 50, "crowd")])
def test_transition_many_users_lds_ranges(native, engine, U, T, tc, kind):
    """k_transition_big (more than 4 096 users: bucket hash in LDS, the row cut into ranges of source tiles): one pass
    (random walk: few destinations per source tile), several passes (users scattered uniformly: every (source,
    destination) pair its own bucket; 1001 tiles), crowded tiles (thousands of users on a handful of tiles: buckets with
    thousands of users); against the literal dict walk of the reference."""
    rng = np.random.default_rng(U + tc)
    if kind == "walk":
        mu, mv = video(U, T, seed=U + T, p_absent=0.05)
    elif kind == "uniform":
        mu = rng.random((T, U))
        mv = np.clip(np.arccos(1.0 - 2.0 * rng.random((T, U))) / np.pi, 0.0, 1.0)
        mu[rng.random((T, U)) < 0.05] = np.nan
    else:
        mu = rng.choice(np.array([0.1, 0.12, 0.5, 0.52, 0.9]), (T, U))
        mv = rng.choice(np.array([0.3, 0.32, 0.7]), (T, U))
    mu[:, 0] = np.where(np.isnan(mu[:, 0]), 0.5, mu[:, 0])
    mv[:, 0] = np.where(np.isnan(mv[:, 0]), 0.5, mv[:, 0])
    plan = plan_for(native, engine, [tc])
    res = plan.transition(mu=mu, mv=mv, want_srccount=True)
    ent, pairs = vo.transition_series(mu, mv, 100, 200, [tc], closed_form=False)
    assert np.array_equal(res["pairs"], pairs)
    n = 2 * (tc // 2) + 1
    src = np.stack([np.bincount(pairs[r][pairs[r][:, 0] >= 0, 0], minlength=n) for r in range(T - 1)])
    assert np.array_equal(res["srccount"], src)
    np.testing.assert_allclose(res["entropy"], ent, rtol=1e-9, equal_nan=True)
    plan.close()


def test_transition_big_srccount_when_bucket_bound_is_a_multiple_of_cap(native, engine):
    """k_transition_big cuts a row's source tiles into ranges of bucket bound cap = 8192 * 6 / 10 - n.  When the row's
    bound sum min(m, n) is a non-zero exact multiple of cap, the empty tiles behind the last populated one belong to no
    range: their srccount must still be written (0), whatever the pooled output buffer held before."""
    tc, W, H = 200, 100, 200
    n = 2 * (tc // 2) + 1
    cap = 8192 * 6 // 10 - n
    tiles = vo.fibonacci_lattice(tc)
    near = vo.nearest_tile(vo.direction_grid(W, H).reshape(-1, 3), tiles)           # [(H+1)(W+1)]
    # one pixel per source tile for the first tiles of the lattice; the last tiles of the lattice stay empty
    pix = {}
    for d, t in enumerate(near):
        pix.setdefault(int(t), d)
    full, rest = divmod(cap, n)
    src_tiles = [t for t in sorted(pix) if t < n - 20][: full + 1]
    assert len(src_tiles) == full + 1
    counts = [n] * full + [rest]
    U, T = 5000, 3
    assert sum(counts) == cap and cap < U
    mu = np.full((T, U), np.nan)
    mv = np.full((T, U), np.nan)
    u = 0
    for t, m in zip(src_tiles, counts):
        d = pix[t]
        px, py = d % (W + 1), d // (W + 1)
        mu[:, u:u + m] = min((px + 0.25) / W, 1.0)
        mv[:, u:u + m] = min((py + 0.25) / H, 1.0)
        u += m
    plan = plan_for(native, engine, [tc])
    # dirty the pooled srccount buffer first: every tile populated
    rng = np.random.default_rng(5)
    dirty_mu = rng.random((T, U))
    dirty_mv = np.clip(np.arccos(1.0 - 2.0 * rng.random((T, U))) / np.pi, 0.0, 1.0)
    dirty = plan.transition(mu=dirty_mu, mv=dirty_mv, want_srccount=True)
    assert (dirty["srccount"] > 0).all()
    res = plan.transition(mu=mu, mv=mv, want_srccount=True)
    ent, pairs = vo.transition_series(mu, mv, W, H, [tc], closed_form=False)
    assert np.array_equal(res["pairs"], pairs)
    src = np.stack([np.bincount(pairs[r][pairs[r][:, 0] >= 0, 0], minlength=n) for r in range(T - 1)])
    assert src.sum(axis=1).tolist() == [cap] * (T - 1) and (src[:, -20:] == 0).all()
    assert np.minimum(src, n).sum(axis=1).tolist() == [cap] * (T - 1)
    assert np.array_equal(res["srccount"], src)
    np.testing.assert_allclose(res["entropy"], ent, rtol=1e-9, equal_nan=True)
    plan.close()


@pytest.mark.parametrize("policy", [1, -1])
def test_weight_rows_of_a_resident_result(native, engine, policy):
    """Weighted spatial results keep the samples' direction ids and compute the weight rows of a fetched block by the
    weights-only pass of the precise sweep: same bits as the eager weights output, any block boundaries, also after the
    plan has been destroyed (the result shares the plan's direction table and lattice-0 tiles); (mu, mv) and ids input."""
    mu, mv = video(70, 90, seed=12)
    plan = plan_for(native, engine, [100, 20], policy=policy)
    eager = plan.spatial(mu=mu, mv=mv, want_weights=True)
    lazy = plan.spatial_resident(mu=mu, mv=mv)
    assert np.array_equal(lazy["entropy"], eager["entropy"], equal_nan=True)
    _, _, weights = vo.spatial_series(mu, mv, 100, 200, [100, 20], want_weights=True)
    np.testing.assert_allclose(eager["weights"], weights, rtol=W_RTOL, atol=w_atol(70))
    plan.close()
    res = lazy["result"]
    assert np.array_equal(res.rows(1, 0, 90), eager["weights"])
    assert np.array_equal(res.rows(1, 37, 5), eager["weights"][37:42])
    assert np.array_equal(res.rows(1, 89, 1), eager["weights"][89:])
    assert np.array_equal(res.rows(0, 0, 90), eager["assign"])
    res.close()
    # explicit direction table (the operator boundary's *_ids entry points)
    rng = np.random.default_rng(3)
    table = vo.vector_from_spherical(np.round(rng.uniform(-180, 180, 200), 1), np.round(rng.uniform(-90, 90, 200), 1))
    ids = rng.integers(-1, 200, (30, 17)).astype(np.int32)
    ids[:, 0] = np.abs(ids[:, 0])
    plan = native.Plan(engine, [vo.fibonacci_lattice(50)], 120.0, 2.0, True, dir_table=table)
    plan.set_table_policy(policy)
    eager = plan.spatial(ids=ids, want_weights=True)
    lazy = plan.spatial_resident(ids=ids)
    plan.close()
    assert np.array_equal(lazy["result"].rows(1, 3, 20), eager["weights"][3:23])
    for t in (0, 29):
        _, hist, _ = vo.spatial_entropy_frame(table[ids[t][ids[t] >= 0]], vo.fibonacci_lattice(50))
        np.testing.assert_allclose(eager["weights"][t], hist, rtol=W_RTOL, atol=w_atol(17))


def test_resident_result_fetched_from_two_threads(native, engine):
    """a lazily computed block of weight rows is staged in ONE buffer per result; two host threads reading
    different blocks of the same result (two cells of one DataFrame) take turns on it and both get the eager bits."""
    import threading
    mu, mv = video(48, 400, seed=31)
    plan = plan_for(native, engine, [100], policy=1)
    eager = plan.spatial(mu=mu, mv=mv, want_weights=True)
    res = plan.spatial_resident(mu=mu, mv=mv)["result"]
    errors = []

    def reader(offset):
        try:
            for it in range(60):
                r0 = (offset + 37 * it) % 380
                n = 1 + (it * 7 + offset) % 20                       # blocks of different sizes: the staging buffer regrows
                if not np.array_equal(res.rows(1, r0, n), eager["weights"][r0:r0 + n]):
                    errors.append(("weights", offset, r0, n))
                if not np.array_equal(res.rows(0, r0, n), eager["assign"][r0:r0 + n]):
                    errors.append(("assign", offset, r0, n))
        except Exception as e:  # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=reader, args=(o,)) for o in (0, 11, 23)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
    assert not errors, errors[:3]
    res.close()
    plan.close()


def test_resident_result_stores_weights_for_wide_audiences(native, engine):
    """direction ids [T][U] i32 are kept instead of the weight rows [T][n_0] f64 only when they are not larger;
    900 users on 21 tiles store the weight rows (the ids would be 21 x their size) — same bits as the eager output either way."""
    mu, mv = video(900, 12, seed=5)
    plan = plan_for(native, engine, [20], policy=1)
    eager = plan.spatial(mu=mu, mv=mv, want_weights=True)
    lazy = plan.spatial_resident(mu=mu, mv=mv)
    assert np.array_equal(lazy["result"].rows(1, 0, 12), eager["weights"])
    assert np.array_equal(lazy["result"].rows(1, 5, 3), eager["weights"][5:8])
    assert np.array_equal(lazy["result"].rows(0, 0, 12), eager["assign"])
    _, _, weights = vo.spatial_series(mu, mv, 100, 200, [20], want_weights=True)
    np.testing.assert_allclose(eager["weights"], weights, rtol=W_RTOL, atol=w_atol(900))
    plan.close()


def _expected_table_rows(dirs):
    """Rows of a weight table = classes of equal Vectors (value equality, -0.0 == 0.0), minus those whose mirror image
    (x, -y, -z) belongs to a class that appears earlier (ensure_alias, vet_plan.hip)."""
    d = np.ascontiguousarray(dirs.reshape(-1, 3) + 0.0)
    first = {}
    for i, row in enumerate(map(bytes, d.view(np.uint8).reshape(len(d), 24))):
        first.setdefault(row, i)
    m = np.ascontiguousarray(d * np.array([1.0, -1.0, -1.0]) + 0.0)
    rows = 0
    for key, i in first.items():
        j = first.get(bytes(m[i].view(np.uint8)))
        rows += 0 if (j is not None and j < i) else 1
    return rows


@pytest.mark.parametrize("W,H", [(100, 200), (6, 4), (200, 400), (640, 480)])
def test_alias_table_built_on_the_device(native, engine, W, H):
    """Round 6: the alias table (equal Vectors and mirror images share a table row) is built by device kernels instead of a
    host hash map; the number of rows is the host rule's, and the gather over those rows agrees with the oracle."""
    plan = plan_for(native, engine, [50], W=W, H=H, policy=1)
    mu, mv = video(40, 25, seed=W)
    res = plan.spatial(mu=mu, mv=mv)
    assert plan.last_formulation(0) == "table"
    assert plan.table_rows() == _expected_table_rows(vo.direction_grid(W, H))
    ent, assign, _ = vo.spatial_series(mu, mv, W, H, [50])
    assert np.array_equal(res["assign"], assign)
    np.testing.assert_allclose(res["entropy"], ent, rtol=1e-8)
    plan.close()
    # no mirror sharing for plans that are not weighted tables of symmetric lattices: an explicit direction table with
    # repeated and mirrored Vectors, -0.0 and +0.0 included
    rng = np.random.default_rng(W)
    base = vo.vector_from_spherical(np.round(rng.uniform(-180, 180, 300), 1), np.round(rng.uniform(-90, 90, 300), 1))
    table = np.concatenate([base, base[::3] * np.array([1.0, -1.0, -1.0]), base[5:40], np.array([[1.0, -0.0, 0.0], [1.0, 0.0, -0.0]])])
    plan = native.Plan(engine, [vo.fibonacci_lattice(50)], 120.0, 2.0, True, dir_table=table)
    plan.set_table_policy(1)
    ids = rng.integers(0, len(table), (12, 33)).astype(np.int32)
    res = plan.spatial(ids=ids)
    assert plan.table_rows() == _expected_table_rows(table)
    for t in (0, 11):
        e, _, _ = vo.spatial_entropy_frame(table[ids[t]], vo.fibonacci_lattice(50))
        np.testing.assert_allclose(res["entropy"][t], e, rtol=1e-8)
    plan.close()


def _weights_fallback_worker(q, env):
    """Child process: weights output and fetched weight rows with / without the exact weight rows (knob read at engine creation)."""
    import os
    os.environ.update(env)
    from viewport_entropy_toolkit import _native
    eng = _native.Engine(0)
    out = {}
    for name, tcs, U, T, policy in (("k2", [100, 20], 70, 40, 1), ("k1", [500], 130, 12, -1)):
        mu, mv = video(U, T, seed=U + T)
        plan = _native.Plan(eng, [vo.fibonacci_lattice(tc) for tc in tcs], 120.0, 2.0, True, 100, 200)
        plan.set_table_policy(policy)
        eager = plan.spatial(mu=mu, mv=mv, want_weights=True)
        lazy = plan.spatial_resident(mu=mu, mv=mv)
        out[name] = (eager["weights"], lazy["result"].rows(1, 3, T - 5), eager["entropy"])
        plan.close()
    q.put(out)


def test_weights_pass_without_the_exact_rows():
    """Plans whose exact weight rows do not fit the device run the precise sweep in weights-only mode instead of
    k_weights_gather (VET_NO_EXACT_ROWS=1 forces that path): the same values (exact weights; the sweep sums strictly in
    column order, the gather per quarter of the users: the last bits may differ), eager == fetched, entropy untouched."""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    res = []
    for e in ({}, {"VET_NO_EXACT_ROWS": "1"}):
        q = ctx.Queue()
        p = ctx.Process(target=_weights_fallback_worker, args=(q, e))
        p.start()
        res.append(q.get(timeout=600))
        p.join(60)
        assert p.exitcode == 0
    for name, tcs, U, T in (("k2", [100, 20], 70, 40), ("k1", [500], 130, 12)):
        a, b = res[0][name], res[1][name]
        mu, mv = video(U, T, seed=U + T)
        _, _, weights = vo.spatial_series(mu, mv, 100, 200, tcs, want_weights=True)
        for r in (a, b):
            np.testing.assert_allclose(r[0], weights, rtol=W_RTOL, atol=w_atol(U))
            assert np.array_equal(r[1], r[0][3:T - 2])                       # fetched block == eager rows, either path
            assert np.array_equal((r[0] > 0) | np.signbit(r[0]), weights > 0)
        np.testing.assert_allclose(b[0], a[0], rtol=1e-13, atol=0)
        assert np.array_equal(a[2], b[2], equal_nan=True)


def test_heavily_clustered_users(native, engine):
    """Everybody looks at the same few tiles: same-address LDS atomics, buckets with many users."""
    rng = np.random.default_rng(1)
    U, T = 512, 40
    mu = np.clip(0.5 + rng.normal(0, 0.004, (T, U)), 0, 1)
    mv = np.clip(0.5 + rng.normal(0, 0.004, (T, U)), 0, 1)
    for weighted, policy in ((True, 1), (True, -1), (False, 0)):
        plan = plan_for(native, engine, [50, 500], weighted=weighted, policy=policy)
        res = plan.spatial(mu=mu, mv=mv)
        ent, assign, _ = vo.spatial_series(mu, mv, 100, 200, [50, 500], use_weight_distribution=weighted)
        assert np.array_equal(res["assign"], assign)
        np.testing.assert_allclose(res["entropy"], ent, rtol=1e-8)
        plan.close()
    plan = plan_for(native, engine, [50, 500])
    tr = plan.transition(mu=mu, mv=mv)
    ent, pairs = vo.transition_series(mu, mv, 100, 200, [50, 500], closed_form=False)
    assert np.array_equal(tr["pairs"], pairs)
    np.testing.assert_allclose(tr["entropy"], ent, rtol=1e-9)
    plan.close()


def test_table_formulation_on_a_larger_grid(native, engine):
    """640x480 pixels: 308 k directions; the direction weight table (118 MB) on a 400 k-sample video."""
    mu, mv = video(1000, 400, seed=99, p_absent=0.02)
    plan = plan_for(native, engine, [50], 640, 480, policy=1)
    res = plan.spatial(mu=mu, mv=mv)
    assert plan.table_stride(0) > 0
    ent, assign, _ = vo.spatial_series(mu, mv, 640, 480, [50])
    assert np.array_equal(res["assign"], assign)
    np.testing.assert_allclose(res["entropy"], ent, rtol=1e-8)
    plan.close()


@pytest.mark.parametrize("fov,power", [(180.0, 2.0), (360.0, 1.0), (0.5, 2.0), (120.0, 50.0), (120.0, 0.01),
                                       (119.999, 2.0), (120.001, 2.0), (60.0, 1.0), (300.0, 0.3)])
@pytest.mark.parametrize("policy", [-1, 1])
def test_extreme_entropy_configs(native, engine, fov, power, policy):
    """Cone edges (fov 180 / 360 / sub-degree), the fast-acos boundary at 120 degrees, and very
    large / small power factors, in both weighted formulations."""
    mu, mv = video(60, 40, seed=int(fov * 10 + power))
    plan = plan_for(native, engine, [100, 20], policy=policy, fov=fov, power=power)
    res = plan.spatial(mu=mu, mv=mv, want_weights=True)
    ent, assign, weights = vo.spatial_series(mu, mv, 100, 200, [100, 20], fov_angle=fov, power_factor=power,
                                             want_weights=True)
    assert np.array_equal(res["assign"], assign)
    # tile_weights values are the reference's whatever the formulation (tests/_tol.py)
    np.testing.assert_allclose(res["weights"], weights, rtol=W_RTOL, atol=w_atol(60, power))
    assert np.array_equal(np.isnan(res["entropy"]), np.isnan(ent))
    ok = np.isfinite(ent)
    fp = "ftable" in (plan.last_formulation(0), plan.last_formulation(1))       # FP32 table weights: |dH|/H <= 1.2e-7
    np.testing.assert_allclose(res["entropy"][ok], ent[ok], rtol=2e-7 if fp else 1e-8, atol=1e-15)
    plan.close()


def test_cabi_argument_validation(native, engine):
    """Bad arguments come back as VET_ERR_INVALID / UNSUPPORTED with a message, not as a crash."""
    import ctypes as C
    lib = native.load_library()
    L = vo.fibonacci_lattice(20)
    with pytest.raises(native.NativeError) as ei:
        native.Plan(engine, [L], 0.0, 2.0, True, 100, 200)              # fov outside (0, 360]
    assert ei.value.code == native.VET_ERR_INVALID and "FOV" in str(ei.value)
    with pytest.raises(native.NativeError):
        native.Plan(engine, [L], 120.0, -1.0, True, 100, 200)           # power <= 0
    with pytest.raises(native.NativeError):
        native.Plan(engine, [np.zeros((4, 3))], 120.0, 2.0, True, 100, 200)   # zero-length tile vector
    plan = native.Plan(engine, [L], 120.0, 2.0, True, 100, 200)
    assert lib.vet_spatial_entropy(plan.handle, None, None, 4, 4, None, None, None, None, None, None) == native.VET_ERR_INVALID
    assert b"NULL" in lib.vet_last_error()
    assert lib.vet_spatial_entropy_host(plan.handle, None, None, None, 4, 4, None, None, None, None) == native.VET_ERR_INVALID
    e = np.empty(4)
    assert lib.vet_spatial_entropy_host(plan.handle, None, None, None, 0, 4, e.ctypes.data_as(C.c_void_p), None, None, None) == native.VET_ERR_INVALID
    # ids beyond the direction table are flagged like out-of-range samples
    res = plan.spatial(ids=np.array([[0, 5, 10 ** 7]], dtype=np.int32), check=False)
    assert res["code"] == native.VET_ERR_RANGE
    plan.close()


@pytest.mark.parametrize("policy", [-1, 1])
def test_very_large_lattice(native, engine, policy):
    """tile_count = 5000 (5001 tiles): more tile groups than waves in the sweep, long table rows."""
    mu, mv = video(50, 12, seed=5)
    plan = plan_for(native, engine, [5000], policy=policy)
    res = plan.spatial(mu=mu, mv=mv, want_weights=True)
    ent, assign, weights = vo.spatial_series(mu, mv, 100, 200, [5000], want_weights=True)
    assert np.array_equal(res["assign"], assign)
    np.testing.assert_allclose(res["entropy"], ent, rtol=1e-8)
    np.testing.assert_allclose(res["weights"], weights, rtol=W_RTOL, atol=w_atol(50))
    tr = plan.transition(mu=mu, mv=mv)
    e2, pairs = vo.transition_series(mu, mv, 100, 200, [5000])
    assert np.array_equal(tr["pairs"], pairs)
    np.testing.assert_allclose(tr["entropy"], e2, rtol=1e-9, equal_nan=True)
    plan.close()


@pytest.mark.parametrize("policy", [-1, 1])
def test_ids_with_several_lattices(native, engine, policy):
    rng = np.random.default_rng(12)
    table = vo.vector_from_spherical(np.round(rng.uniform(-180, 180, 500), 1), np.round(rng.uniform(-90, 90, 500), 1))
    ids = rng.integers(0, 500, (30, 70)).astype(np.int32)
    ids[rng.random(ids.shape) < 0.2] = -1
    ids[:, 3] = 7
    tcs = [100, 20, 250]
    plan = native.Plan(engine, [vo.fibonacci_lattice(tc) for tc in tcs], 120.0, 2.0, True, dir_table=table)
    plan.set_table_policy(policy)
    res = plan.spatial(ids=ids)
    ref = np.zeros(len(ids))
    for tc in tcs:
        L = vo.fibonacci_lattice(tc)
        ref += np.array([vo.spatial_entropy_frame(table[r[r >= 0]], L)[0] for r in ids])
    np.testing.assert_allclose(res["entropy"], ref / len(tcs), rtol=1e-8)
    assert np.array_equal(res["present"], (ids >= 0).sum(1))
    plan.close()


@pytest.mark.parametrize("policy", [1, -1, 0])
def test_batch_of_videos_in_one_launch(native, engine, policy):
    """vet_spatial_entropy_batch: videos of different shapes share one launch (table formulation) or
    run one by one inside the call (other policies); results equal the per-video calls bit for bit."""
    shapes = [(8, 30), (64, 100), (33, 7), (300, 12), (1, 5), (64, 100)]
    vids = [video(u, t, seed=10 * i + u) for i, (u, t) in enumerate(shapes)]
    plan = plan_for(native, engine, [50, 100], policy=policy)
    got = plan.spatial_batch(vids, want_assign=True)
    for (mu, mv), g in zip(vids, got):
        one = plan.spatial(mu=mu, mv=mv)
        if policy != 0:                # under 'auto' the formulation may switch between the calls
            assert np.array_equal(g["entropy"], one["entropy"])
        assert np.array_equal(g["assign"], one["assign"]) and np.array_equal(g["present"], one["present"])
        ent, assign, _ = vo.spatial_series(mu, mv, 100, 200, [50, 100])
        np.testing.assert_allclose(g["entropy"], ent, rtol=1e-8)
    plan.close()


def _fused_variants_worker(q, env):
    """Child process: the same plans through the fused table's other kernels, selected by environment."""
    import os
    os.environ.update(env)
    from viewport_entropy_toolkit import _native, _synthetic
    eng = _native.Engine(0)
    out = {}
    for name, tcs, U, T in FUSED_CASES:
        mu, mv = _synthetic.random_walk_video(U, T, base_seed=17, p_absent=0.1)
        plan = _native.Plan(eng, [vo.fibonacci_lattice(tc) for tc in tcs], 120.0, 2.0, True, 100, 200)
        plan.set_table_policy(1)
        plan.set_raw_weights(True)               # the tables' own histograms are what the variants are compared on
        r = plan.spatial(mu=mu, mv=mv, want_weights=True)
        out[name] = (r["entropy"], r["assign"], r["weights"], plan.last_formulation(0))
        plan.close()
    q.put(out)


FUSED_CASES = (("k1", [500], 96, 700), ("k3", [50, 100, 200], 200, 900), ("k5", [20, 50, 100, 250, 1000], 64, 300))


@pytest.mark.parametrize("env", [{"VET_FUSED": "1"}, {"VET_NO_FUSED": "1"}])
def test_fused_table_kernels_agree(env):
    """The fused table (one row per distinct direction over all lattices; VET_FUSED=1: also for one-lattice plans) and
    the per-lattice tables (VET_NO_FUSED=1) agree with each other and with the oracle.  (The knobs are read once, when
    the engine is created: each variant runs in a process of its own.)"""
    import multiprocessing as mp
    from viewport_entropy_toolkit import _synthetic
    ctx = mp.get_context("spawn")
    res = []
    for e in ({}, env):
        q = ctx.Queue()
        p = ctx.Process(target=_fused_variants_worker, args=(q, e))
        p.start()
        res.append(q.get(timeout=600))
        p.join(60)
        assert p.exitcode == 0
    for name, tcs, U, T in FUSED_CASES:
        a, b = res[0][name], res[1][name]
        assert a[3] == b[3] == "table"
        assert np.array_equal(a[1], b[1])
        if "VET_NO_FUSED" in env or name == "k1":
            # per-lattice rows carry their own block-floating-point shift, fused rows a shared one: different roundings
            np.testing.assert_allclose(b[0], a[0], rtol=1e-9, equal_nan=True)
        else:
            assert np.array_equal(a[0], b[0], equal_nan=True), name          # the same fused rows either way
        np.testing.assert_allclose(b[2], a[2], rtol=0, atol=2.0 ** -33 * U)
        mu, mv = _synthetic.random_walk_video(U, T, base_seed=17, p_absent=0.1)
        ent, assign, _ = vo.spatial_series(mu[:60], mv[:60], 100, 200, tcs)
        assert np.array_equal(b[1][:60], assign)
        np.testing.assert_allclose(b[0][:60], ent, rtol=1e-8, equal_nan=True)


@pytest.mark.parametrize("tcs,shapes", [([50], [(8, 30), (64, 100), (34, 7), (300, 12), (2, 5), (64, 100)]),
                                        ([50, 100, 200], [(64, 60), (33, 41), (1, 9), (128, 30)]),        # an odd user count: single-user loads
                                        ([20, 50], [(4096, 3), (10, 400)])])
def test_unweighted_batch_in_one_launch(native, engine, tcs, shapes):
    """vet_spatial_entropy_batch with use_weight_distribution=False: every video's frame blocks in one k_spatial_u_lds
    launch per lattice (several lattices: k_finalize_batch); integer counts and table logarithms, so the results equal
    the per-video calls bit for bit, and the oracle."""
    vids = [video(u, t, seed=7 * i + u) for i, (u, t) in enumerate(shapes)]
    plan = plan_for(native, engine, tcs, weighted=False)
    got = plan.spatial_batch(vids, want_assign=True, check=False)
    for (mu, mv), g in zip(vids, got):
        one = plan.spatial(mu=mu, mv=mv, check=False)
        assert np.array_equal(g["entropy"], one["entropy"], equal_nan=True)
        assert np.array_equal(g["assign"], one["assign"]) and np.array_equal(g["present"], one["present"])
        if mu.shape[0] * mu.shape[1] <= 8000:
            ent, assign, _ = vo.spatial_series(mu, mv, 100, 200, tcs, use_weight_distribution=False)
            np.testing.assert_allclose(g["entropy"], ent, rtol=1e-9, equal_nan=True)
    plan.close()


@pytest.mark.parametrize("tcs,shapes", [([200], [(8, 30), (64, 100), (33, 7), (300, 12), (2, 5), (64, 100)]),
                                        ([20, 50], [(64, 60), (512, 41), (100, 9)]),
                                        ([50], [(64, 50)] * 9)])
def test_transition_batch_in_one_launch(native, engine, tcs, shapes):
    """vet_transition_entropy_batch: every video with its own workgroups in one k_transition_run launch per lattice;
    pairs and counts equal the per-video calls exactly, the entropies to the summation order of the cell sums
    (the workgroup size follows the largest video), and the literal dict-walk oracle."""
    vids = []
    for i, (u, t) in enumerate(shapes):
        mu, mv = video(u, t, seed=3 * i + u, p_absent=0.05)
        mu[:, 0] = np.where(np.isnan(mu[:, 0]), 0.5, mu[:, 0])          # user 0 is in every frame: no empty frame pair
        mv[:, 0] = np.where(np.isnan(mv[:, 0]), 0.5, mv[:, 0])
        vids.append((mu, mv))
    plan = plan_for(native, engine, tcs)
    got = plan.transition_batch(vids, want_pairs=True)
    for (mu, mv), g in zip(vids, got):
        one = plan.transition(mu=mu, mv=mv)
        assert np.array_equal(g["pairs"], one["pairs"]) and np.array_equal(g["common"], one["common"])
        np.testing.assert_allclose(g["entropy"], one["entropy"], rtol=1e-12, equal_nan=True)
        ent, pairs = vo.transition_series(mu, mv, 100, 200, tcs, closed_form=False)
        assert np.array_equal(g["pairs"], pairs)
        np.testing.assert_allclose(g["entropy"], ent, rtol=1e-9, equal_nan=True)
    plan.close()


@pytest.mark.parametrize("tcs,fov,power", [([50, 100], 120.0, 20.0), ([500], 10.0, 2.0), ([50], 120.0, 150.0), ([20, 50], 60.0, 200.0)])
def test_batch_with_fp_tables_and_marker_plans(native, engine, tcs, fov, power):
    """vet_spatial_entropy_batch on plans that take the FP table: one launch for plain FP tables (sorted rows, per-wave
    histograms: bit-identical to the per-video calls), video by video where the table holds marker entries (their frames
    go through the in-call resolver); NaN frames as the oracle has them."""
    shapes = [(8, 30), (200, 40), (33, 7), (300, 12), (1, 5), (130, 25)]
    vids = [video(u, t, seed=5 * i + u) for i, (u, t) in enumerate(shapes)]
    plan = plan_for(native, engine, tcs, policy=1, fov=fov, power=power)
    plan.spatial_batch(vids, want_assign=True, check=False)          # builds the tables
    engine.profile_enable(True)
    engine.profile_reset()
    got = plan.spatial_batch(vids, want_assign=True, check=False)
    launches = engine.profile_get("k_spatial")[1]
    engine.profile_enable(False)
    assert plan.last_formulation(0) == "ftable"
    if power <= 30.0:
        # no weight below 2^-1048: in-FoV weights below FP32 range are stored as the smallest subnormal, the table holds no
        # marker, nothing needs the resolver and the whole batch is ONE launch
        assert launches == 1, launches
    else:
        assert launches >= len(vids)                                  # marker plans: video by video, plus their resolvers
    for (mu, mv), g in zip(vids, got):
        one = plan.spatial(mu=mu, mv=mv, check=False)
        assert np.array_equal(g["entropy"], one["entropy"], equal_nan=True)
        assert np.array_equal(g["assign"], one["assign"]) and np.array_equal(g["present"], one["present"])
        ent, assign, _ = vo.spatial_series(mu, mv, 100, 200, tcs, fov_angle=fov, power_factor=power)
        assert np.array_equal(np.isnan(g["entropy"]), np.isnan(ent))
        ok = ~np.isnan(ent)
        np.testing.assert_allclose(g["entropy"][ok], ent[ok], rtol=1e-6, atol=1e-15)
    plan.close()


@pytest.mark.parametrize("U,T,tcs,fov,power", [(3000, 7, [50], 120.0, 20.0), (2500, 5, [50, 100], 120.0, 30.0), (3000, 6, [500], 120.0, 150.0)])
def test_fp_table_with_more_users_than_one_chunk(native, engine, U, T, tcs, fov, power):
    """FP table on frames of more than 2048 users: the users arrive in several chunks (set and histograms side by side in
    LDS, every chunk's row list rank-sorted), marker plans hand their undecided frames to the resolver, which walks the
    users in chunks of 1024 too."""
    mu, mv = video(U, T, seed=U + T)
    plan = plan_for(native, engine, tcs, policy=1, fov=fov, power=power)
    a = plan.spatial(mu=mu, mv=mv, want_weights=True, check=False)
    assert plan.last_formulation(0) == "ftable"
    b = plan.spatial(mu=mu, mv=mv, check=False)
    assert np.array_equal(a["entropy"], b["entropy"], equal_nan=True)
    ent, assign, weights = vo.spatial_series(mu, mv, 100, 200, tcs, fov_angle=fov, power_factor=power, want_weights=True)
    assert np.array_equal(a["assign"], assign)
    assert np.array_equal(np.isnan(a["entropy"]), np.isnan(ent))
    ok = ~np.isnan(ent)
    np.testing.assert_allclose(a["entropy"][ok], ent[ok], rtol=1e-6, atol=1e-15)
    assert np.array_equal((a["weights"] != 0) | np.signbit(a["weights"]), (weights != 0) | np.signbit(weights))
    plan.close()

