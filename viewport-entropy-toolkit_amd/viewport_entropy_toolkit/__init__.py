"""MI355X-native viewport-entropy engine behind the viewport-entropy-toolkit API.

Import surface of the reference package for the spatial / transition entropy path
(reference __init__.py:7-26).  The compute runs in hand-written HIP kernels reached through
a ctypes C-ABI (``_native``); nothing here falls back to a CPU implementation.
"""

from .data_types import Point, RadialPoint, Vector, ValidationError, SpatialError, convert_vectors_to_coordinates
from .config import AnalyzerConfig, DEFAULT_VIDEO_DIMENSIONS, DEFAULT_TILE_COUNTS
from .analyzers import SpatialEntropyAnalyzer, TransitionEntropyAnalyzer, NaiveSpatialEntropyAnalyzer

__version__ = "1.0.0"
__all__ = [
    "Point", "RadialPoint", "Vector", "ValidationError", "SpatialError", "convert_vectors_to_coordinates",
    "AnalyzerConfig", "SpatialEntropyAnalyzer", "TransitionEntropyAnalyzer", "NaiveSpatialEntropyAnalyzer",
    "DEFAULT_VIDEO_DIMENSIONS", "DEFAULT_TILE_COUNTS",
]
