"""Utilities of the entropy path.

The operator-level functions keep the reference's names and signatures and run on the HIP
engine; the ingest helpers are vectorised host code; visualisation is reduced to the
configuration object and the entropy-over-time graph (rendering is outside the engine).
"""

from . import data_utils as _data, entropy_utils as _entropy, visualization_utils as _viz

_PUBLIC = {
    _data: ("generate_fibonacci_lattice", "normalize_to_pixel", "pixel_to_spherical", "process_viewport_data",
            "format_trajectory_data", "validate_video_dimensions", "get_fb_tile_boundaries", "get_lat_lon_tiles",
            "normalize", "find_perpendicular_on_tangent_plane", "great_circle_intersection", "get_line_segment",
            "find_nearest_point", "spherical_interpolation", "get_tile_corners", "triangulate_spherical_polygon",
            "angle_at_vertex", "calculate_spherical_triangle_area", "compute_spherical_polygon_area",
            "compute_fb_tile_areas", "compute_lat_lon_tile_areas"),
    _entropy: ("EntropyConfig", "vector_angle_distance", "find_angular_distances", "find_nearest_tile", "calculate_tile_weights", "compute_spatial_entropy",
               "compute_transition_entropy", "calculate_naive_tile_weights", "find_naive_tile_index",
               "compute_naive_spatial_entropy"),
    _viz: ("VisualizationConfig", "save_graph"),
}
for _module, _names in _PUBLIC.items():
    for _name in _names:
        globals()[_name] = getattr(_module, _name)

__all__ = [name for names in _PUBLIC.values() for name in names]
del _module, _names, _name
