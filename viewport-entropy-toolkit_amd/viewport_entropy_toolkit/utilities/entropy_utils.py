"""Entropy configuration and the operator-level entry points of the reference
(utilities/entropy_utils.py:20-38, 89-144, 147-332), served by the HIP engine.

Callers pass arbitrary ``Vector`` objects here, so each call builds a small explicit
direction table and runs the same kernels as the analyzers through the ``*_ids`` C-ABI
entry points.  For whole videos use the analyzers: they keep the device tables alive.
"""

from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Tuple

import numpy as np

from ..data_types import RadialPoint, Vector, ValidationError
from .. import _native


@dataclass
class EntropyConfig:
    """Field-of-view weighting parameters."""

    fov_angle: float = 120.0
    use_weight_distribution: bool = True
    power_factor: float = 2.0

    def __post_init__(self) -> None:
        if not 0 < self.fov_angle <= 360:
            raise ValidationError("FOV angle must be between 0 and 360 degrees")
        if self.power_factor <= 0:
            raise ValidationError("Power factor must be positive")


def _xyz(vectors) -> np.ndarray:
    return np.array([[v.x, v.y, v.z] for v in vectors], dtype=np.float64).reshape(-1, 3)


def _plan_for(vectors: List[Vector], tile_centers: List[Vector], config: EntropyConfig) -> "_native.Plan":
    plan = _native.Plan(_native.Engine.default(), [_xyz(tile_centers)], config.fov_angle, config.power_factor,
                        config.use_weight_distribution, dir_table=_xyz(vectors))
    plan.set_table_policy(-1)        # a handful of samples: sweep directly, full 2^-52 weight resolution
    return plan


def vector_angle_distance(v1: Vector, v2: Vector) -> float:
    """Angle between two vectors in radians: arccos(clip(dot(v1/|v1|, v2/|v2|), -1, 1)), on the HIP engine.

    As in the reference, anything that goes wrong (a non-Vector argument, no device) surfaces as
    ``ValidationError("Error calculating vector angle: ...")``; a zero-length vector gives nan."""
    try:
        pair = np.array([[v1.x, v1.y, v1.z], [v2.x, v2.y, v2.z]], dtype=np.float64)
        return np.float64(_native.Engine.default().angular_distances(pair[:1], pair[1:])[0, 0])
    except _native.NativeUnavailable:
        raise                                   # no HIP extension / no device: fail loudly, there is no CPU path
    except Exception as e:
        raise ValidationError(f"Error calculating vector angle: {str(e)}")


def find_angular_distances(vector: Vector, tile_centers: List[Vector]) -> np.ndarray:
    """``[n, 2]`` array of ``[tile index, angular distance]`` rows (one kernel launch for the whole row)."""
    if len(tile_centers) == 0:
        return np.array([])
    try:
        d = _native.Engine.default().angular_distances(_xyz([vector]), _xyz(tile_centers))[0]
    except _native.NativeUnavailable:
        raise
    except Exception as e:
        raise ValidationError(f"Error calculating vector angle: {str(e)}")
    return np.column_stack([np.arange(len(tile_centers), dtype=np.float64), d])


def find_nearest_tile(vector: Vector, tile_centers: List[Vector]) -> int:
    """Index of the tile centre at the smallest angular distance (lowest index on ties)."""
    plan = _plan_for([vector], tile_centers, EntropyConfig())
    try:
        return int(plan.read_nearest(0)[0])
    finally:
        plan.close()


def calculate_tile_weights(vector: Vector, tile_centers: List[Vector], config: EntropyConfig) -> Dict[Vector, float]:
    """FoV weights of one viewing direction: ((max - d) / max) ** power for tiles closer than
    fov/2, or weight 1.0 on the nearest tile when the distribution is switched off."""
    plan = _plan_for([vector], tile_centers, config)
    try:
        if not config.use_weight_distribution:
            return {tile_centers[int(plan.read_nearest(0)[0])]: 1.0}
        row = plan.spatial(ids=np.zeros((1, 1), dtype=np.int32), want_assign=False, want_weights=True)["weights"][0]
        # -0.0 = in the FoV with a weight that underflowed to 0.0: still a key of the reference's dict
        return {tile_centers[int(i)]: float(row[i]) + 0.0 for i in np.argsort(-row, kind="stable") if row[i] > 0 or np.signbit(row[i])}
    finally:
        plan.close()


def compute_spatial_entropy(vector_dict: Dict[str, Vector], tile_centers: List[Vector],
                            config: EntropyConfig) -> Tuple[float, Dict[Vector, float], Dict[str, int]]:
    """Normalised Shannon entropy of one frame's tile weight distribution.

    Returns (entropy, {tile Vector: summed weight}, {identifier: nearest tile index})."""
    if not vector_dict:
        raise ValidationError("Empty vector dictionary")
    if not tile_centers:
        raise ValidationError("No tile centers provided")
    users = [(k, v) for k, v in vector_dict.items() if v is not None]
    if not users:
        if config.use_weight_distribution:
            return 0.0 / _native._quantiser.max_entropy(len(tile_centers)), {}, {}
        raise ZeroDivisionError("float division by zero")
    plan = _plan_for([v for _, v in users], tile_centers, config)
    try:
        ids = np.arange(len(users), dtype=np.int32)[None, :]
        res = plan.spatial(ids=ids, want_assign=True, want_weights=True)
    finally:
        plan.close()
    row = res["weights"][0]
    weights = {tile_centers[int(i)]: float(row[i]) + 0.0 for i in np.nonzero((row > 0) | np.signbit(row))[0]}
    assignments = {k: int(a) for (k, _), a in zip(users, res["assign"][0])}
    return float(res["entropy"][0]), weights, assignments


def compute_transition_entropy(prior_vector_dict: dict, current_vector_dict: dict, tile_centers: List[Vector],
                               config: EntropyConfig, FOV_angle: float) -> Tuple[float, Dict[Vector, int], Dict[str, Tuple[int, int]]]:
    """Normalised transition entropy between two frames over users present in both.

    ``config`` and ``FOV_angle`` do not influence the value (as in the reference).
    Returns (entropy, {source tile Vector: user count}, {identifier: (prior idx, current idx)})."""
    if not prior_vector_dict or not current_vector_dict:
        raise ValidationError("Empty vector dictionary")
    if not tile_centers:
        raise ValidationError("No tile centers provided")
    users = [k for k in current_vector_dict if k in prior_vector_dict]
    if not users:
        raise ZeroDivisionError("division by zero")          # the reference's `1 / total_weight` with the int 0 (:326)
    vecs = [prior_vector_dict[k] for k in users] + [current_vector_dict[k] for k in users]
    plan = _plan_for(vecs, tile_centers, EntropyConfig())
    try:
        U = len(users)
        ids = np.arange(2 * U, dtype=np.int32).reshape(2, U)
        res = plan.transition(ids=ids, want_pairs=True, want_srccount=True)
    finally:
        plan.close()
    src = res["srccount"][0]
    weights = {tile_centers[int(i)]: int(src[i]) for i in np.nonzero(src > 0)[0]}
    assignments = {k: (int(p), int(c)) for k, (p, c) in zip(users, res["pairs"][0])}
    return float(res["entropy"][0]), weights, assignments


# --------------------------------------------------------------------------------------------
# naive latitude-longitude grid tiling (reference utilities/entropy_utils.py:335-452)
# --------------------------------------------------------------------------------------------
def find_naive_tile_index(point: RadialPoint, tile_height: float, tile_width: float) -> str:
    """``"{lon index}_{lat index}"`` of the grid cell the point sits in (truncating division)."""
    return f"{int((point.lon + 180) / tile_width)}_{int((point.lat + 90) / tile_height)}"


def calculate_naive_tile_weights(point: RadialPoint, tile_height: float, tile_width: float,
                                 config: EntropyConfig) -> Dict[str, float]:
    """The whole weight 1.0 goes to the point's own grid cell."""
    return {find_naive_tile_index(point, tile_height, tile_width): 1.0}


def naive_tile_count(tile_height, tile_width) -> int:
    return int(180.0 / tile_height) * int(360.0 / tile_width)


def compute_naive_spatial_entropy(points_dict: Dict[str, RadialPoint], tile_height: int, tile_width: int,
                                  config: EntropyConfig) -> Tuple[float, Dict[str, float], Dict[str, str]]:
    """Normalised Shannon entropy of one frame's users over the lat/lon grid cells.

    Returns (entropy, {cell key: user count}, {identifier: cell key}).  The cell index arithmetic
    is the quantiser and runs on the host; the histogram and entropy run on the HIP engine."""
    if not points_dict:
        raise ValidationError("Empty radial points dictionary")
    if not tile_height or not tile_width:
        raise ValidationError("No tile dimensions provided")
    if 180 % tile_height != 0:
        raise ValidationError("Tile height must divide 180!")
    if 360 % tile_width != 0:
        raise ValidationError("Tile width must divide 360!")
    users = [(k, find_naive_tile_index(p, tile_height, tile_width)) for k, p in points_dict.items() if p is not None]
    if not users:
        raise ZeroDivisionError("float division by zero")
    cells = sorted({c for _, c in users})
    cell_id = {c: i for i, c in enumerate(cells)}
    num_tiles = naive_tile_count(tile_height, tile_width)
    plan = _native.Plan(_native.Engine.default(), [None], config.fov_angle, config.power_factor,
                        config.use_weight_distribution, dir_table=np.tile([1.0, 0.0, 0.0], (len(users), 1)),
                        bin_luts=[np.array([cell_id[c] for _, c in users], dtype=np.uint16)],
                        bin_counts=[len(cells)], bin_max_entropy=[_native._quantiser.max_entropy(num_tiles)],
                        bin_norm_tiles=[num_tiles])
    try:
        res = plan.spatial(ids=np.arange(len(users), dtype=np.int32)[None, :], want_assign=False, want_weights=True)
    finally:
        plan.close()
    weights = {c: float(res["weights"][0][i]) for c, i in cell_id.items()}
    return float(res["entropy"][0]), weights, {k: c for k, c in users}
