"""Utilities of the entropy path (names as in the reference's ``utilities`` package)."""

from .data_utils import (
    generate_fibonacci_lattice,
    normalize_to_pixel,
    pixel_to_spherical,
    process_viewport_data,
    format_trajectory_data,
    validate_video_dimensions,
)
from .entropy_utils import (
    EntropyConfig,
    find_nearest_tile,
    calculate_tile_weights,
    compute_spatial_entropy,
    compute_transition_entropy,
    calculate_naive_tile_weights,
    find_naive_tile_index,
    compute_naive_spatial_entropy,
)
from .visualization_utils import VisualizationConfig, save_graph

__all__ = [
    "generate_fibonacci_lattice", "normalize_to_pixel", "pixel_to_spherical", "process_viewport_data",
    "format_trajectory_data", "validate_video_dimensions",
    "EntropyConfig", "find_nearest_tile", "calculate_tile_weights", "compute_spatial_entropy",
    "compute_transition_entropy", "calculate_naive_tile_weights", "find_naive_tile_index",
    "compute_naive_spatial_entropy",
    "VisualizationConfig", "save_graph",
]
