"""Tile boundary / area geometry (reference utilities/data_utils.py:58-189, 412-741).
CPU part: the oracle restatement and the package's host-side corner walk / spherical areas against goldens made
by the live reference (tests/golden/g11_geometry.npz, oracle/gen_golden.py G11).
GPU part (marked): get_fb_tile_boundaries / compute_fb_tile_areas through the HIP kernel k_fb_boundaries."""
import numpy as np
import pytest

from oracle import vet_oracle as vo
from viewport_entropy_toolkit import Vector
from viewport_entropy_toolkit.data_types import ValidationError
from viewport_entropy_toolkit.utilities import (calculate_spherical_triangle_area, compute_fb_tile_areas,
                                                compute_lat_lon_tile_areas, compute_spherical_polygon_area,
                                                find_nearest_point, get_fb_tile_boundaries, get_lat_lon_tiles,
                                                get_tile_corners, great_circle_intersection, spherical_interpolation,
                                                triangulate_spherical_polygon)

TCS = (20, 50, 100, 33)


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(golden_dir / "g11_geometry.npz")


def golden_edges(g, tc, i):
    return [[Vector(*g[f"tc{tc}__edges"][i, e, 0]), Vector(*g[f"tc{tc}__edges"][i, e, 1])]
            for e in range(int(g[f"tc{tc}__edge_count"][i]))]


@pytest.mark.parametrize("tc", TCS)
def test_oracle_boundaries_and_areas_match_the_reference(g, tc):
    edges = vo.fb_tile_boundaries(tc)
    assert np.array_equal([len(e) for e in edges], g[f"tc{tc}__edge_count"])
    for i, lst in enumerate(edges):
        for e, (p1, p2) in enumerate(lst):
            np.testing.assert_allclose(p1, g[f"tc{tc}__edges"][i, e, 0], rtol=0, atol=1e-14)
            np.testing.assert_allclose(p2, g[f"tc{tc}__edges"][i, e, 1], rtol=0, atol=1e-14)
        c = vo.tile_corners(lst)
        assert len(c) == g[f"tc{tc}__corner_count"][i]
        np.testing.assert_allclose(c, g[f"tc{tc}__corners"][i, :len(c)], rtol=0, atol=1e-12)
    areas, frac = vo.fb_tile_areas(tc)
    np.testing.assert_allclose(areas, g[f"tc{tc}__areas"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(frac, g[f"tc{tc}__fractions"], rtol=0, atol=1e-13)


@pytest.mark.parametrize("tc", TCS)
def test_host_side_corner_walk_and_areas_on_reference_edges(g, tc):
    """get_tile_corners / compute_spherical_polygon_area of the package on the reference's own edges."""
    n = len(g[f"tc{tc}__edge_count"])
    for i in range(n):
        edges = golden_edges(g, tc, i)
        corners = get_tile_corners(edges)
        want = g[f"tc{tc}__corners"][i, :g[f"tc{tc}__corner_count"][i]]
        assert len(corners) == len(want)
        np.testing.assert_allclose([[c.x, c.y, c.z] for c in corners], want, rtol=0, atol=1e-12)
        assert len(triangulate_spherical_polygon(corners)) == len(corners) - 2
        assert abs(compute_spherical_polygon_area(edges) - g[f"tc{tc}__areas"][i]) < 1e-12


def test_small_helpers_match_the_reference(g):
    assert calculate_spherical_triangle_area(Vector(1.0, 0.0, 0.0), Vector(0.0, 1.0, 0.0), Vector(0.0, 0.0, 1.0)) == g["octant_area"]
    got = [calculate_spherical_triangle_area(*[Vector(*p) for p in t]) for t in g["tri_points"]]
    np.testing.assert_allclose(got, g["tri_areas"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(got, [vo.spherical_triangle_area(*t) for t in g["tri_points"]], rtol=0, atol=1e-13)
    for a, b, p in zip(g["gc_n1"], g["gc_n2"], g["gc_p1"]):
        p1, p2 = great_circle_intersection(a, b)
        np.testing.assert_allclose(p1, p, rtol=0, atol=1e-15)
        np.testing.assert_allclose(p2, -p, rtol=0, atol=1e-15)
    a, b = Vector(1.0, 0.0, 0.0), Vector(0.0, 1.0, 0.0)
    np.testing.assert_allclose(spherical_interpolation(a, b, 0.5), [np.sqrt(0.5), np.sqrt(0.5), 0.0], atol=1e-15)
    assert find_nearest_point(a, b, Vector(0.9, 0.1, 0.0)) is a and find_nearest_point(a, b, Vector(0.5, 0.5, 0.0)) is b
    with pytest.raises(ValueError):
        triangulate_spherical_polygon([a, b])
    with pytest.raises(ValidationError):
        compute_lat_lon_tile_areas(0, 4)
    with pytest.raises(ValidationError):
        compute_fb_tile_areas(0)
    with pytest.raises(ValidationError):
        get_fb_tile_boundaries(-1)


def test_lat_lon_tiling_covers_the_sphere():
    tiles = get_lat_lon_tiles(8, 4)
    assert len(tiles) == 32 and len(tiles["0_0"]) == 3 and len(tiles["1_0"]) == 4 and len(tiles["3_7"]) == 3
    areas, frac = compute_lat_lon_tile_areas(8, 4)
    assert abs(sum(frac.values()) - 1.0) < 1e-12 and abs(sum(areas.values()) - 4 * np.pi) < 1e-11


# --------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("tc", TCS)
def test_hip_boundaries_match_the_reference(g, tc):
    b = get_fb_tile_boundaries(tc)
    n = len(g[f"tc{tc}__edge_count"])
    assert sorted(b) == list(range(n))
    assert np.array_equal([len(b[i]) for i in range(n)], g[f"tc{tc}__edge_count"])
    for i in range(n):
        for e, (p1, p2) in enumerate(b[i]):
            np.testing.assert_allclose([p1.x, p1.y, p1.z], g[f"tc{tc}__edges"][i, e, 0], rtol=0, atol=1e-12)
            np.testing.assert_allclose([p2.x, p2.y, p2.z], g[f"tc{tc}__edges"][i, e, 1], rtol=0, atol=1e-12)
    areas, frac = compute_fb_tile_areas(tc)
    np.testing.assert_allclose([areas[i] for i in range(n)], g[f"tc{tc}__areas"], rtol=0, atol=1e-9)
    np.testing.assert_allclose([frac[i] for i in range(n)], g[f"tc{tc}__fractions"], rtol=0, atol=1e-10)


@pytest.mark.gpu
@pytest.mark.parametrize("tc", [1, 2, 5, 250, 1000])
def test_hip_boundaries_match_the_oracle_on_other_lattices(tc):
    from viewport_entropy_toolkit import _native, _quantiser
    edges, count = _native.Engine.default().fb_tile_boundaries(_quantiser.lattice_xyz(tc))
    ref = vo.fb_tile_boundaries(tc)
    assert np.array_equal(count, [len(e) for e in ref])
    for i, lst in enumerate(ref):
        for e, (p1, p2) in enumerate(lst):
            np.testing.assert_allclose(edges[i, e, 0], p1, rtol=0, atol=1e-12)
            np.testing.assert_allclose(edges[i, e, 1], p2, rtol=0, atol=1e-12)
        assert np.isnan(edges[i, len(lst):]).all()
    if tc >= 250:
        areas, frac = compute_fb_tile_areas(tc)
        assert abs(sum(frac.values()) - 1.0) < 1e-9
        ra, _ = vo.fb_tile_areas(tc)
        np.testing.assert_allclose([areas[i] for i in range(len(ra))], ra, rtol=0, atol=1e-9)
