"""Tile boundary and area geometry of the sphere tilings (reference utilities/data_utils.py:58-225, 412-741;
SURVEY.md §8f-4).

``get_fb_tile_boundaries`` — the neighbour / bisector search over all tile pairs, O(n^2) Python loops per tiling
in the reference — runs on the HIP engine (kernel ``k_fb_boundaries`` behind ``vet_fb_tile_boundaries``); the
corner walk and the spherical-excess areas on top of its edges are small host code with the reference's
semantics (4-decimal corner keys, fan triangulation from the first corner).  Same names, arguments, return
types and exceptions as the reference.
"""

from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np

from .. import _native, _quantiser
from ..data_types import ValidationError, Vector

_CORNER_DECIMALS = 4


def _xyz(v) -> np.ndarray:
    return np.array([v.x, v.y, v.z], dtype=np.float64)


def _as_vector(a) -> Vector:
    return Vector(float(a[0]), float(a[1]), float(a[2]))


# ------------------------------------------------------------------ small vector helpers (data_utils.py:412-528)
def normalize(v: np.ndarray) -> np.ndarray:
    """``v`` scaled to unit length."""
    return v / np.linalg.norm(v)


def get_line_segment(v1: Vector, v2: Vector) -> np.ndarray:
    """Chord from ``v2`` to ``v1`` as an array."""
    return _xyz(v1) - _xyz(v2)


def find_perpendicular_on_tangent_plane(vec: np.ndarray, midpoint: np.ndarray) -> np.ndarray:
    """Unit vector perpendicular to ``vec`` in the plane tangent to the sphere at ``midpoint``."""
    return normalize(np.cross(normalize(midpoint), vec))


def great_circle_intersection(n1: np.ndarray, n2: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """The two antipodal points where the great circles with normals ``n1`` and ``n2`` meet."""
    p = normalize(np.cross(normalize(n1), normalize(n2)))
    return p, -p


def find_nearest_point(v1: Vector, v2: Vector, compare_vector: Vector) -> Vector:
    """Whichever of ``v1`` / ``v2`` has the shorter chord to ``compare_vector`` after rounding the chords to
    4 decimals; ``v2`` on a tie."""
    d1 = np.linalg.norm(get_line_segment(compare_vector, v1)).round(4)
    d2 = np.linalg.norm(get_line_segment(compare_vector, v2)).round(4)
    return v1 if d1 < d2 else v2


def spherical_interpolation(v1: Vector, v2: Vector, t: float) -> np.ndarray:
    """Slerp between the directions of ``v1`` and ``v2`` along the shorter arc."""
    a, b = normalize(_xyz(v1)), normalize(_xyz(v2))
    theta = np.arccos(np.clip(np.dot(a, b), -1.0, 1.0))
    return (np.sin((1 - t) * theta) * a + np.sin(t * theta) * b) / np.sin(theta)


# ------------------------------------------------------------------ tilings
def get_fb_tile_boundaries(tile_count: int) -> Dict[int, List[List[Vector]]]:
    """{tile index: [[P1, P2], ...]} — the boundary edges of every tile of the Fibonacci tiling, from the engine."""
    if tile_count <= 0:
        raise ValidationError("Tile counts cannot be less than 1 for to visualize tiling!")
    tiles = _quantiser.lattice_xyz(tile_count)
    edges, count = _native.Engine.default().fb_tile_boundaries(tiles)
    return {i: [[_as_vector(edges[i, e, 0]), _as_vector(edges[i, e, 1])] for e in range(int(count[i]))]
            for i in range(len(tiles))}


def get_lat_lon_tiles(num_tiles_horizontal: int, num_tiles_vertical: int, radius: float = 1.0) -> Dict[str, List[List[Vector]]]:
    """{"row_col": edges} of a latitude / longitude tiling; the rows touching a pole are triangles."""
    dlat, dlon = 180 / num_tiles_vertical, 360 / num_tiles_horizontal
    north, south = Vector(0, 0, radius), Vector(0, 0, -radius)
    tiles: Dict[str, List[List[Vector]]] = {}
    for i in range(num_tiles_vertical):
        lo_lat, hi_lat = -90 + i * dlat, -90 + (i + 1) * dlat
        for j in range(num_tiles_horizontal):
            west, east = -180 + j * dlon, -180 + (j + 1) * dlon
            sw, se = Vector.from_spherical(lat=lo_lat, lon=west), Vector.from_spherical(lat=lo_lat, lon=east)
            ne, nw = Vector.from_spherical(lat=hi_lat, lon=east), Vector.from_spherical(lat=hi_lat, lon=west)
            if hi_lat >= 90:
                tiles[f"{i}_{j}"] = [[sw, se], [sw, north], [se, north]]
            elif lo_lat <= -90:
                tiles[f"{i}_{j}"] = [[ne, nw], [ne, south], [nw, south]]
            else:
                tiles[f"{i}_{j}"] = [[sw, se], [sw, nw], [se, ne], [ne, nw]]
    return tiles


# ------------------------------------------------------------------ corners and areas
def get_tile_corners(tile_boundaries: List[List[Vector]]) -> List[Vector]:
    """Corners of one tile in edge order (consecutive corners share an edge), keyed on coordinates rounded to
    4 decimals; starts with the two ends of the first edge and walks on until no unvisited neighbour is left."""
    key = lambda p: p.round(decimals=_CORNER_DECIMALS)      # noqa: E731
    walk = [key(tile_boundaries[0][0]), key(tile_boundaries[0][1])]
    visited = {walk[0]: True, walk[1]: True}
    links: Dict[Vector, List[Vector]] = {}
    for a, b in tile_boundaries:
        ka, kb = key(a), key(b)
        links.setdefault(ka, []).append(kb)
        links.setdefault(kb, []).append(ka)
        visited.setdefault(ka, False)
        visited.setdefault(kb, False)
    here = walk[1]
    while not visited[links[here][0]] or not visited[links[here][1]]:
        here = links[here][0] if not visited[links[here][0]] else links[here][1]
        walk.append(here)
        visited[here] = True
    return walk


def triangulate_spherical_polygon(tile_corners: List[Vector]) -> List[List[Vector]]:
    """Fan of triangles anchored at the first corner."""
    if len(tile_corners) < 3:
        raise ValueError("At least 3 boundary points are needed for a polygon.")
    return [[tile_corners[0], tile_corners[i], tile_corners[i + 1]] for i in range(1, len(tile_corners) - 1)]


def angle_at_vertex(v1: np.ndarray, v2: np.ndarray, v3: np.ndarray) -> float:
    """Angle at ``v1`` between the great-circle arcs towards ``v2`` and ``v3`` (unit vectors)."""
    t2 = v2 - np.dot(v2, v1) * v1
    t3 = v3 - np.dot(v3, v1) * v1
    t2 = t2 / np.linalg.norm(t2)
    t3 = t3 / np.linalg.norm(t3)
    return np.arccos(np.clip(np.dot(t2, t3), -1.0, 1.0))


def calculate_spherical_triangle_area(P1: Vector, P2: Vector, P3: Vector, radius: float = 1.0) -> float:
    """Spherical excess of the triangle times ``radius``^2."""
    a, b, c = (normalize(_xyz(p)) for p in (P1, P2, P3))
    excess = angle_at_vertex(a, b, c) + angle_at_vertex(b, c, a) + angle_at_vertex(c, a, b) - np.pi
    return excess * (radius ** 2)


def compute_spherical_polygon_area(tile_boundaries: List[List[Vector]], radius=1.0) -> float:
    """Area of the tile bounded by ``tile_boundaries``: sum over the fan triangles of its corner walk."""
    total = 0.0
    for tri in triangulate_spherical_polygon(get_tile_corners(tile_boundaries)):
        total += calculate_spherical_triangle_area(tri[0], tri[1], tri[2], radius)
    return total


def _areas_of(boundaries: dict):
    areas = {k: compute_spherical_polygon_area(edges) for k, edges in boundaries.items()}
    whole = 4 * np.pi
    return areas, {k: a / whole for k, a in areas.items()}


def compute_fb_tile_areas(tile_count: int) -> Tuple[Dict[int, float], Dict[int, float]]:
    """({tile: area}, {tile: fraction of the sphere}) of the Fibonacci tiling."""
    if tile_count <= 0:
        raise ValidationError("Number of points must be positive!")
    return _areas_of(get_fb_tile_boundaries(tile_count))


def compute_lat_lon_tile_areas(num_tiles_horizontal: int, num_tiles_vertical: int) -> Tuple[Dict[str, float], Dict[str, float]]:
    """({tile: area}, {tile: fraction of the sphere}) of the latitude / longitude tiling."""
    if num_tiles_horizontal <= 0 or num_tiles_vertical <= 0:
        raise ValidationError("Number of tiles horizontal and vertical must be positive!")
    return _areas_of(get_lat_lon_tiles(num_tiles_horizontal, num_tiles_vertical))
