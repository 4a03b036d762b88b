"""Pins the CPU oracle (oracle/vet_oracle.py) against golden vectors produced by
the real reference (oracle/gen_golden.py).  Tile indices bit-exact; entropy and
weights within 1e-9 relative (the contract with the HIP path is 1e-6)."""
import numpy as np
import pytest

from oracle import vet_oracle as vo

RTOL = 1e-9


def load(golden_dir, name):
    return np.load(golden_dir / name, allow_pickle=False)


def test_g1_lattices(golden_dir):
    g = load(golden_dir, "g1_lattices.npz")
    for key in g.files:
        tc = int(key[2:])
        L = vo.fibonacci_lattice(tc)
        assert L.shape == g[key].shape == (2 * (tc // 2) + 1, 3)
        assert np.array_equal(L, g[key]), key


@pytest.mark.parametrize("W,H", [(100, 200), (3840, 1920), (6, 4)])
def test_g2_axis_tables(golden_dir, W, H):
    g = load(golden_dir, "g2_quantiser.npz")
    lon, lat = vo.axis_tables(W, H)
    assert np.array_equal(lon, g[f"lon_{W}x{H}"])
    assert np.array_equal(lat, g[f"lat_{W}x{H}"])
    # the remap quirk: -180 -> 0.0 and -90 -> 0.0
    assert lon[0] == 0.0 and lat[H] == 0.0


def test_g2_vector_grid(golden_dir):
    g = load(golden_dir, "g2_quantiser.npz")
    assert np.array_equal(vo.direction_grid(100, 200), g["vec_100x200"])
    lon, lat = vo.axis_tables(3840, 1920)
    v = vo.vector_from_spherical(lon[g["big_px"]], lat[g["big_py"]])
    assert np.array_equal(v, g["big_vec"])


def test_g3_nearest_exhaustive(golden_dir):
    p = golden_dir / "g3_nearest.npz"
    g = np.load(p)
    flat = vo.direction_grid(100, 200).reshape(-1, 3)
    for key in g.files:
        tc = int(key[2:])
        near = vo.nearest_tile(flat, vo.fibonacci_lattice(tc)).reshape(201, 101)
        assert np.array_equal(near, g[key]), key


def test_g7_weight_rows(golden_dir):
    g = load(golden_dir, "g7_weight_rows.npz")
    grid = vo.direction_grid(100, 200)
    dirs = grid[g["py"], g["px"]]
    for tag, tc, kw in (("tc500", 500, {}),
                        ("tc50_p15_fov90", 50, dict(power_factor=1.5, fov_angle=90.0)),
                        ("tc100_fov360", 100, dict(fov_angle=360.0, power_factor=3.0))):
        L = vo.fibonacci_lattice(tc)
        rows = vo.tile_weight_rows(dirs, L, **kw)
        ref = g[f"{tag}__rows"]
        assert np.array_equal(rows > 0, ref > 0), tag
        np.testing.assert_allclose(rows, ref, rtol=1e-9, atol=1e-15)
        assert np.array_equal(vo.nearest_tile(dirs, L), g[f"{tag}__nearest"])


def _dense_from_g4(g):
    tracks = [(g["time_in"][u], g["mu_in"][u], g["mv_in"][u]) for u in range(len(g["mu_in"]))]
    return tracks


@pytest.mark.parametrize("tag,tcs,kw", [
    ("w_tc50", [50], {}),
    ("w_tc50_100_200", [50, 100, 200], {}),
    ("u_tc50", [50], dict(use_weight_distribution=False)),
    ("u_tc20_50", [20, 50], dict(use_weight_distribution=False)),
    ("w_tc50_p15", [50], dict(power_factor=1.5)),
    ("w_tc50_fov90", [50], dict(fov_angle=90.0)),
    ("w_tc100_fov200_p05", [100], dict(fov_angle=200.0, power_factor=0.5)),
])
def test_g4_spatial_analyzer(golden_dir, tag, tcs, kw):
    g = load(golden_dir, "g4_spatial.npz")
    cols = [str(c) for c in g[f"{tag}__columns"]]
    order = [int(c[4:]) for c in cols]           # reference column (glob) order
    tracks = [_dense_from_g4(g)[u] for u in order]
    times, mu, mv = vo.format_trajectories(tracks)
    assert np.array_equal(times, g[f"{tag}__time"])
    ent, assign, weights = vo.spatial_series(mu, mv, 100, 200, tcs, want_weights=True, **kw)
    assert np.array_equal(assign, g[f"{tag}__assign"])
    np.testing.assert_allclose(ent, g[f"{tag}__entropy"], rtol=RTOL, equal_nan=True)
    fr = g[f"{tag}__weights_frames"]
    np.testing.assert_allclose(weights[fr], g[f"{tag}__weights"], rtol=RTOL, atol=1e-15)


@pytest.mark.parametrize("tag,tcs", [("tc200", [200]), ("tc20_50", [20, 50])])
@pytest.mark.parametrize("closed", [False, True])
def test_g5_transition_analyzer(golden_dir, tag, tcs, closed):
    g = load(golden_dir, "g5_transition.npz")
    cols = [str(c) for c in g[f"{tag}__columns"]]
    order = [int(c[4:]) for c in cols]
    tracks = [_dense_from_g4(g)[u] for u in order]
    times, mu, mv = vo.format_trajectories(tracks)
    assert np.array_equal(times[1:], g[f"{tag}__time"])
    ent, pairs = vo.transition_series(mu, mv, 100, 200, tcs, closed_form=closed)
    assert np.array_equal(pairs, g[f"{tag}__pairs"])
    np.testing.assert_allclose(ent, g[f"{tag}__entropy"], rtol=RTOL, equal_nan=True)


def test_g6_ingest_edges(golden_dir):
    g = load(golden_dir, "g6_ingest.npz")
    for tag, kw in (("w", {}), ("u", dict(use_weight_distribution=False))):
        cols = [str(c) for c in g[f"{tag}__columns"]]
        tracks = [tuple(g[f"in_{c}"][:, i] for i in range(3)) for c in cols]
        times, mu, mv = vo.format_trajectories(tracks)
        assert np.array_equal(times, g[f"{tag}__time"])
        px, py, present, grid = vo.sample_directions(mu, mv, 100, 200)
        lon_axis, lat_axis = vo.axis_tables(100, 200)
        ref_ll, ref_xyz = g[f"{tag}__lonlat"], g[f"{tag}__xyz"]
        assert np.array_equal(present, ~np.isnan(ref_ll[..., 0]))
        assert np.array_equal(lon_axis[px[present]], ref_ll[..., 0][present])
        assert np.array_equal(lat_axis[py[present]], ref_ll[..., 1][present])
        assert np.array_equal(grid[py[present], px[present]], ref_xyz[present])
        ent, assign, _ = vo.spatial_series(mu, mv, 100, 200, [50, 20], **kw)
        assert np.array_equal(assign, g[f"{tag}__assign"])
        np.testing.assert_allclose(ent, g[f"{tag}__entropy"], rtol=RTOL, equal_nan=True)
    assert str(g["t__raised"]) == "ZeroDivisionError"
    with pytest.raises(ZeroDivisionError):
        vo.transition_series(mu, mv, 100, 200, [50])


@pytest.mark.parametrize("tc", [20, 50])
def test_g8_dense_transition(golden_dir, tc):
    g = load(golden_dir, "g8_dense_transition.npz")
    grid = vo.direction_grid(100, 200)
    L = vo.fibonacci_lattice(tc)
    px, py, present = g["px"], g["py"], g["present"]
    near = vo.nearest_tile(grid.reshape(-1, 3), L).reshape(201, 101)
    T = px.shape[0]
    for t in range(1, T):
        both = present[t] & present[t - 1]
        p = near[py[t - 1][both], px[t - 1][both]]
        c = near[py[t][both], px[t][both]]
        assert np.array_equal(np.stack([p, c], 1), g[f"tc{tc}__pairs"][t - 1][both])
        ref = g[f"tc{tc}__entropy"][t - 1]
        for fn in (vo.transition_entropy_pairs, vo.transition_entropy_closed_form):
            np.testing.assert_allclose(fn(p, c, len(L)), ref, rtol=RTOL, equal_nan=True)
        src = np.bincount(p, minlength=len(L))
        assert np.array_equal(src, g[f"tc{tc}__srccount"][t - 1])
    for tag, uw in (("u", False), ("w", True)):
        for t in range(T):
            d = grid[py[t][present[t]], px[t][present[t]]]
            e, _, _ = vo.spatial_entropy_frame(d, L, use_weight_distribution=uw)
            np.testing.assert_allclose(e, g[f"tc{tc}__spatial_{tag}"][t], rtol=RTOL, equal_nan=True)


def test_g9_operator_edges(golden_dir):
    g = load(golden_dir, "g9_operator_edges.npz")
    L = vo.fibonacci_lattice(50)
    v, v2 = g["v"], g["v2"]
    e, _, _ = vo.spatial_entropy_frame(v[None], L, use_weight_distribution=False)
    assert np.isnan(e) and np.isnan(g["single_unweighted"])
    e, _, _ = vo.spatial_entropy_frame(v[None], L)
    np.testing.assert_allclose(e, g["single_weighted"], rtol=RTOL)
    e, hist, near = vo.spatial_entropy_frame(np.stack([v, v2]), L, fov_angle=1.0)
    assert e == 0.0 == float(g["tiny_fov"]) and int(g["tiny_fov_nweights"]) == 0 and hist.sum() == 0
    assert np.array_equal(near, g["tiny_fov_assign"])
    e, _, _ = vo.spatial_entropy_frame(np.stack([v, v2]), vo.fibonacci_lattice(1))
    assert np.isnan(e) and np.isnan(g["one_tile"])
    nv, nv2 = vo.nearest_tile(v[None], L)[0], vo.nearest_tile(v2[None], L)[0]
    for fn in (vo.transition_entropy_pairs, vo.transition_entropy_closed_form):
        assert np.isnan(fn([nv], [nv2], len(L))) and np.isnan(g["trans_single"])
        np.testing.assert_allclose(fn([nv, nv2], [nv2, nv], len(L)), g["trans_two"], rtol=RTOL)
        with pytest.raises(ZeroDivisionError):
            fn([], [], len(L))
    assert str(g["spatial_empty"]) == "ValidationError"
    assert str(g["trans_disjoint"]) == "ZeroDivisionError"


@pytest.mark.parametrize("th,tw", [(10, 10), (30, 45), (20, 20), (90, 180)])
@pytest.mark.parametrize("flag", [True, False])
def test_g10_naive_analyzer(golden_dir, th, tw, flag):
    g = load(golden_dir, "g10_naive.npz")
    tag = f"h{th}_w{tw}_{'w' if flag else 'u'}"
    cols = [str(c) for c in g[f"{tag}__columns"]]
    order = [int(c[4:]) for c in cols]
    tracks = [(g["time_in"][u], g["mu_in"][u], g["mv_in"][u]) for u in order]
    times, mu, mv = vo.format_trajectories(tracks)
    ent, li, lj = vo.naive_series(mu, mv, 100, 200, th, tw, flag)
    assert np.array_equal(times, g[f"{tag}__time"]) and bool(g[f"{tag}__weights_is_none"])
    np.testing.assert_allclose(ent, g[f"{tag}__entropy"], rtol=RTOL, equal_nan=True)
    for fi in (0, 150, 299):
        assert [f"{a}_{b}" for a, b in zip(li[fi], lj[fi])] == [str(x) for x in g[f"{tag}__f{fi}_assign"]]
        keys, counts = np.unique([f"{a}_{b}" for a, b in zip(li[fi], lj[fi])], return_counts=True)
        assert sorted(keys) == [str(k) for k in g[f"{tag}__f{fi}_wkeys"]]
        assert np.array_equal(counts[np.argsort(keys)].astype(float), g[f"{tag}__f{fi}_wvals"])
        np.testing.assert_allclose(ent[fi], g[f"{tag}__f{fi}_entropy"], rtol=RTOL)


G12_CONFIGS = [(tc, fov, power) for tc in (50, 500) for fov in (120, 60) for power in (50, 80, 100, 150, 200)]


def g12_samples(g):
    """G12's pixel coordinates as normalised samples (NaN = absent)."""
    px, py = g["px"], g["py"]
    present = px >= 0
    mu = np.where(present, np.where(px == 100, 1.0, (px + 0.5) / 100.0), np.nan)
    mv = np.where(present, np.where(py == 200, 1.0, (py + 0.5) / 200.0), np.nan)
    return mu, mv


@pytest.mark.parametrize("tc,fov,power", G12_CONFIGS)
def test_g12_underflowed_weights_give_nan(golden_dir, tc, fov, power):
    """FoV weights that underflow to 0.0 stay keys of the reference's dict and make the frame NaN
    (entropy_utils.py:131-135, 195-198): NaN pattern, finite values and the 0.0-valued keys."""
    g = load(golden_dir, "g12_underflow.npz")
    tag = f"tc{tc}_fov{fov}_p{power}"
    mu, mv = g12_samples(g)
    ent, _, w = vo.spatial_series(mu, mv, 100, 200, [tc], fov_angle=float(fov), power_factor=float(power), want_weights=True)
    ref = g[f"{tag}__entropy"]
    assert np.array_equal(np.isnan(ent), np.isnan(ref))
    np.testing.assert_allclose(ent, ref, rtol=RTOL, atol=1e-15, equal_nan=True)
    keys = (w != 0) | np.signbit(w)
    assert np.array_equal(keys, g[f"{tag}__keys"])
    np.testing.assert_allclose(np.abs(w), g[f"{tag}__hist"], rtol=1e-9, atol=0)
    grid = vo.direction_grid(100, 200)
    L = vo.fibonacci_lattice(tc)
    for f in (0, 7, 15, 20, 33):                    # the operator-level restatement too
        ok = g["px"][f] >= 0
        e, _, _ = vo.spatial_entropy_frame(grid[g["py"][f][ok], g["px"][f][ok]], L, float(fov), float(power))
        np.testing.assert_allclose(e, ref[f], rtol=RTOL, atol=1e-15, equal_nan=True)


def test_g13_angular_distances_and_off_grid_nearest(golden_dir):
    """vector_angle_distance / find_angular_distances / find_nearest_tile of the live reference (entropy_utils.py:41-106):
    the oracle's batched restatement gives the same distances (the cosines may differ by an ulp or two between one
    np.dot per pair and a matrix product: |delta| <= 8 * 2^-52 / sin d + 4 ulp) and the same nearest tiles, on the default
    grid and on a 640 x 480 one."""
    g = np.load(golden_dir / "g13_angular.npz")
    for tc in (500, 50, 2):
        tiles = vo.fibonacci_lattice(tc)
        ref = g[f"tc{tc}__dist"]
        got = vo.angular_distances(g["dirs"], tiles)
        bound = 8 * 2.0 ** -52 / np.maximum(np.sin(ref), 1.5e-8) + 4 * np.spacing(ref)
        assert (np.abs(got - ref) <= bound).all()
        assert np.array_equal(vo.nearest_tile(g["dirs"], tiles), g[f"tc{tc}__nearest"])
        assert np.array_equal(np.argmin(ref, axis=1), g[f"tc{tc}__nearest"])
    got = vo.angular_distances(g["raw_a"], g["raw_b"])
    np.testing.assert_allclose(got, g["raw_dist"], rtol=1e-14, atol=3e-8)
    lon, lat = vo.axis_tables(640, 480)
    vec = vo.vector_from_spherical(lon[g["big_px"]], lat[g["big_py"]])
    assert np.array_equal(vec, g["big_vec"])
    for tc in (50, 500):
        assert np.array_equal(vo.nearest_tile(vec, vo.fibonacci_lattice(tc)), g[f"big_tc{tc}__nearest"])
