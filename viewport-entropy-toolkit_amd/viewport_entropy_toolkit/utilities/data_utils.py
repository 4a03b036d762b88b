"""Lattice generation and ingest helpers with the reference's names and semantics
(utilities/data_utils.py:25-56, 227-410).  The tile-boundary / spherical-area geometry of the same reference
module (:58-225, 412-741) lives in ``geometry.py`` and is re-exported here under the reference's names.
"""

from __future__ import annotations

from pathlib import Path
from typing import List, Tuple, Union

import numpy as np
import pandas as pd

from ..data_types import Point, RadialPoint, Vector, ValidationError
from .. import _ingest, _quantiser
from .geometry import (angle_at_vertex, calculate_spherical_triangle_area, compute_fb_tile_areas,  # noqa: F401
                       compute_lat_lon_tile_areas, compute_spherical_polygon_area, find_nearest_point,
                       find_perpendicular_on_tangent_plane, get_fb_tile_boundaries, get_lat_lon_tiles,
                       get_line_segment, get_tile_corners, great_circle_intersection, normalize,
                       spherical_interpolation, triangulate_spherical_polygon)


def generate_fibonacci_lattice(num_points: int) -> List[Vector]:
    """Tile centres as Vectors; yields 2*floor(num_points/2)+1 of them, like the reference."""
    if num_points <= 0:
        raise ValidationError("Number of points must be positive")
    return [Vector(float(x), float(y), float(z)) for x, y, z in _quantiser.lattice_xyz(num_points)]


def validate_video_dimensions(width: int, height: int) -> None:
    _ingest.check_video_dimensions(width, height)


def normalize_to_pixel(normalized: np.ndarray, dimension: int) -> np.ndarray:
    return _ingest.to_pixels(normalized, dimension)


def pixel_to_spherical(point: Point, video_width: int, video_height: int) -> RadialPoint:
    validate_video_dimensions(video_width, video_height)
    if point.pixel_x > video_width or point.pixel_y > video_height:
        raise ValidationError("Pixel coordinates exceed video dimensions")
    return RadialPoint(lon=(point.pixel_x / video_width) * 360 - 180,
                       lat=90 - (point.pixel_y / video_height) * 180)


def process_viewport_data(filepath: Union[str, Path], video_width: int, video_height: int) -> Tuple[pd.DataFrame, str]:
    """CSV (time, 2dmu, 2dmv) -> cleaned DataFrame with pixel and lon/lat columns, and the file stem."""
    return _ingest.read_track(filepath, video_width, video_height)


def format_trajectory_data(trajectory_data: List[Tuple[str, pd.DataFrame]]) -> Tuple[pd.DataFrame, pd.DataFrame]:
    """Per-user tracks -> (points_df, vectors_df): one row per frame (first-appearance order of
    the rounded time), one column per user, cells RadialPoint / Vector / None.

    API-compatible materialisation for callers that want the reference's object frames; the
    analyzers themselves work on the dense arrays of ``_ingest.build_dense``.
    """
    if not trajectory_data:
        raise ValidationError("No trajectory data provided")
    names = [name for name, _ in trajectory_data]
    tracks = []
    for _, data in trajectory_data:
        data["time"] = data["time"].round(1)
        tracks.append((data["time"].to_numpy(dtype=np.float64), data["lon"].to_numpy(dtype=np.float64),
                       data["lat"].to_numpy(dtype=np.float64)))
    times, lon, lat = _ingest.build_dense(tracks)
    points = {"time": list(times)}
    vectors = {"time": list(times)}
    for u, name in enumerate(names):
        pcol: list = [None] * len(times)
        vcol: list = [None] * len(times)
        for f in np.nonzero(~np.isnan(lon[:, u]))[0]:
            lo, la = round(float(lon[f, u]), 1), round(float(lat[f, u]), 1)
            if lo <= -180:
                lo = (lo + 360) % 360 - 180
            if la <= -90:
                la = (la + 180) % 180 - 90
            pcol[f] = RadialPoint(lon=lo, lat=la)
            vcol[f] = Vector.from_spherical(lo, la)
        points[name] = pcol
        vectors[name] = vcol
    return pd.DataFrame(points), pd.DataFrame(vectors)
