"""Value types of the viewport-entropy path: pixel point, lon/lat point, direction vector.

Drop-in for the reference's ``viewport_entropy_toolkit.data_types`` (same class names,
fields, validation and error types; reference data_types.py:20-216).  ``Vector`` is the
hashable tile / direction key the reference uses in its result dictionaries, and
``Vector.from_spherical`` is the quantiser (6-decimal rounding) the HIP kernels reproduce
bit-exactly from host-built axis tables (see ``_quantiser.py``).
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import List, Tuple, Union

import numpy as np

from . import _quantiser


class SpatialError(Exception):
    """Root of the package's error hierarchy."""


class ValidationError(SpatialError):
    """Invalid spatial data or configuration."""


@dataclass(frozen=True)
class Point:
    """Pixel coordinates of a viewport centre (non-negative)."""

    pixel_x: int
    pixel_y: int

    def __post_init__(self) -> None:
        if self.pixel_x < 0 or self.pixel_y < 0:
            raise ValidationError("Pixel coordinates cannot be negative")

    def as_tuple(self) -> Tuple[int, int]:
        return (self.pixel_x, self.pixel_y)


@dataclass(frozen=True)
class RadialPoint:
    """Viewing direction as longitude [-180, 180] and latitude [-90, 90] in degrees."""

    lon: float
    lat: float

    def __post_init__(self) -> None:
        if not -180 <= self.lon <= 180:
            raise ValidationError("Longitude must be between -180 and 180 degrees")
        if not -90 <= self.lat <= 90:
            raise ValidationError("Latitude must be between -90 and 90 degrees")

    def normalize_coordinates(self) -> "RadialPoint":
        return RadialPoint(((self.lon + 180) % 360) - 180, ((self.lat + 90) % 180) - 90)

    def as_tuple(self) -> Tuple[float, float]:
        return (self.lon, self.lat)


@dataclass(frozen=True)
class Vector:
    """Cartesian direction; frozen, so hashable by (x, y, z) and usable as a dict key."""

    x: float
    y: float
    z: float

    def __post_init__(self) -> None:
        if self.length() == 0:
            raise ValidationError("Vector cannot have zero length")

    def length(self) -> float:
        return np.sqrt(self.x ** 2 + self.y ** 2 + self.z ** 2)

    def normalize(self) -> "Vector":
        n = self.length()
        if n == 0:
            raise ValidationError("Cannot normalize zero-length vector")
        return Vector(x=self.x / n, y=self.y / n, z=self.z / n)

    def dot_product(self, other: "Vector") -> float:
        return self.x * other.x + self.y * other.y + self.z * other.z

    def as_tuple(self) -> Tuple[float, float, float]:
        return (self.x, self.y, self.z)

    def round(self, decimals: int) -> "Vector":
        return Vector(x=np.round(self.x, decimals=decimals),
                      y=np.round(self.y, decimals=decimals),
                      z=np.round(self.z, decimals=decimals))

    @classmethod
    def from_spherical(cls, lon: float, lat: float) -> "Vector":
        """Direction of (lon, lat) degrees with each component rounded to 6 decimals."""
        if not -180 <= lon <= 180:
            raise ValidationError("Longitude must be between -180 and 180 degrees")
        if not -90 <= lat <= 90:
            raise ValidationError("Latitude must be between -90 and 90 degrees")
        x, y, z = _quantiser.vector_xyz(lon, lat)
        return cls(x=float(x), y=float(y), z=float(z))


def convert_vectors_to_coordinates(vectors: Union[List[Vector], np.ndarray]) -> Tuple[List[float], List[float]]:
    """Plot helper: (lon, lat) degree lists of a sequence of Vectors (off the hot path)."""
    lons, lats = [], []
    for v in vectors:
        if v is None:
            continue
        u = v.normalize() if isinstance(v, Vector) else Vector(*map(float, v)).normalize()
        lons.append(float(np.degrees(np.arctan2(u.y, u.x))))
        lats.append(float(90.0 - np.degrees(np.arccos(np.clip(u.z, -1.0, 1.0)))))
    return lons, lats
