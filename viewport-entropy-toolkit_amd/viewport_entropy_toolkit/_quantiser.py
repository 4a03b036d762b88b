"""Host-side quantiser tables, built with the same float operations as the reference.

The reference turns a sample into a direction through a chain of roundings:
``px = int(mu*W)`` -> ``lon = px/W*360-180`` -> Python ``round(lon, 1)`` -> the
``<= -180`` remap -> radians -> sin/cos -> numpy ``round(., 6)``
(utilities/data_utils.py:243-286, 390-397; data_types.py:204-216).  A GPU must not redo the
trigonometry or the decimal rounding itself, so the host evaluates them once per *axis*
(W+1 longitudes, H+1 latitudes) and the device only multiplies and applies the exactly
reproducible ``rint(v*1e6)/1e6``.  Lattice centres come from the same code.
"""

from __future__ import annotations

from typing import Tuple

import numpy as np

_PHI = (1 + np.sqrt(5)) / 2


def vector_xyz(lon, lat) -> np.ndarray:
    """Rounded Cartesian components for lon/lat in degrees (scalars or arrays)."""
    lon, lat = np.broadcast_arrays(np.asarray(lon, dtype=np.float64), np.asarray(lat, dtype=np.float64))
    theta = np.radians(lon)
    phi = np.radians(90 - lat)
    sp = np.sin(phi)
    return np.stack([np.round(sp * np.cos(theta), 6), np.round(sp * np.sin(theta), 6),
                     np.round(np.cos(phi), 6)], axis=-1)


def axis_angles(width: int, height: int) -> Tuple[np.ndarray, np.ndarray]:
    """lon(px) for px = 0..W and lat(py) for py = 0..H, after rounding and remap."""
    lon = np.empty(width + 1, dtype=np.float64)
    for px in range(width + 1):
        v = round(float((px / width) * 360 - 180), 1)
        if v <= -180:
            v = (v + 360) % 360 - 180
        lon[px] = v
    lat = np.empty(height + 1, dtype=np.float64)
    for py in range(height + 1):
        v = round(float(90 - (py / height) * 180), 1)
        if v <= -90:
            v = (v + 180) % 180 - 90
        lat[py] = v
    return lon, lat


def axis_trig(width: int, height: int):
    """(cos theta, sin theta)[W+1] and (sin phi, cos phi)[H+1] for the C-ABI plan."""
    lon, lat = axis_angles(width, height)
    theta = np.radians(lon)
    phi = np.radians(90 - lat)
    return (np.ascontiguousarray(np.cos(theta)), np.ascontiguousarray(np.sin(theta)),
            np.ascontiguousarray(np.sin(phi)), np.ascontiguousarray(np.cos(phi)))


def lattice_xyz(tile_count: int) -> np.ndarray:
    """Fibonacci-lattice centres, [2*floor(n/2)+1, 3] (utilities/data_utils.py:25-56)."""
    half = int(tile_count / 2)
    idx = range(-half, half + 1)
    lat = np.array([np.arcsin(2 * i / (2 * half + 1)) * 180 / np.pi for i in idx])
    lon = np.array([(((i % _PHI) * 360 / _PHI) + 180) % 360 - 180 for i in idx])
    return np.ascontiguousarray(vector_xyz(lon, lat))


def max_entropy(n_tiles: int) -> float:
    """-n * (1/n) * log2(1/n), the reference's normaliser (entropy_utils.py:201-203)."""
    p = 1.0 / n_tiles
    with np.errstate(all="ignore"):
        return float(-n_tiles * p * np.log2(p))
