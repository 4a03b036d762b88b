"""NaiveSpatialEntropyAnalyzer: per-frame entropy of the users over a latitude-longitude grid of
``tile_height`` x ``tile_width`` degree cells (reference analyzers/naive_spatial_entropy.py:102-152,
utilities/entropy_utils.py:383-452), on the HIP engine's integer histogram kernels.

The cell of a sample depends only on its pixel direction, so the host turns the two axis tables
(lon per px, lat per py) into a direction -> cell table once and the device path is the same
16-B-in / 4-B-out stream as the unweighted Fibonacci mode.  As in the reference the result frame
carries ``None`` in ``tile_weights`` / ``tile_assignments``."""

from __future__ import annotations

import logging
import time
from typing import Optional

import numpy as np
import pandas as pd

from .. import _native, _quantiser
from ..config import NaiveAnalyzerConfig
from ..data_types import ValidationError
from ..utilities.entropy_utils import naive_tile_count
from ._base import _EntropyAnalyzerBase

logger = logging.getLogger(__name__)


class NaiveSpatialEntropyAnalyzer(_EntropyAnalyzerBase):
    """Drop-in analyzer for the lat/lon-cut tiling; ``config`` is a ``NaiveAnalyzerConfig``."""

    _logger = logger

    def __init__(self, config: Optional[NaiveAnalyzerConfig] = None):
        self.config = config or NaiveAnalyzerConfig()
        self.plot_manager = None
        self._data_cache = {}
        self._entropy_results = None
        self._fibonacci_vectors = {}
        self._dense = None
        self._plan = None
        self._plan_key = None
        self.last_timing = {}

    def _naive_plan(self) -> "_native.Plan":
        cfg = self.config
        th, tw = cfg.tile_height, cfg.tile_width
        if 180 % th != 0:
            raise ValidationError("Tile height must divide 180!")
        if 360 % tw != 0:
            raise ValidationError("Tile width must divide 360!")
        key = (cfg.video_width, cfg.video_height, th, tw, cfg.entropy_config.use_weight_distribution)
        if self._plan is None or self._plan_key != key:
            lon, lat = _quantiser.axis_angles(cfg.video_width, cfg.video_height)
            li = ((lon + 180) / tw).astype(np.int64)            # int() truncation, entropy_utils.py:378-379
            lj = ((lat + 90) / th).astype(np.int64)
            n_lat = int(lj.max()) + 1
            bins = (int(li.max()) + 1) * n_lat
            if li.min() < 0 or lj.min() < 0 or bins > 65535:
                raise ValidationError("tile dimensions give an unsupported number of grid cells")
            lut = (li[None, :] * n_lat + lj[:, None]).astype(np.uint16)      # [H+1][W+1]
            num_tiles = naive_tile_count(th, tw)
            ec = cfg.entropy_config
            self._plan = _native.Plan(_native.Engine.default(), [None], ec.fov_angle, ec.power_factor,
                                      ec.use_weight_distribution, cfg.video_width, cfg.video_height,
                                      bin_luts=[lut], bin_counts=[bins],
                                      bin_max_entropy=[_quantiser.max_entropy(num_tiles)], bin_norm_tiles=[num_tiles])
            self._plan_key = key
        return self._plan

    def compute_entropy(self) -> pd.DataFrame:
        if not self._data_cache or self._dense is None:
            raise ValidationError("No data available. Call process_directory first.")
        times, mu, mv, _ = self._dense
        t_start = time.perf_counter()
        try:
            res = self._naive_plan().spatial(mu=mu, mv=mv, want_assign=False, want_weights=False)
        except _native.NativeError as e:
            if e.code == _native.VET_ERR_RANGE:
                raise ValidationError(str(e))
            if e.code == _native.VET_ERR_EMPTY:
                raise ValidationError("Empty radial points dictionary")
            raise
        self._record_compute(time.perf_counter() - t_start, mu.size, len(times))
        self._entropy_results = pd.DataFrame({
            "time": times,
            "entropy": res["entropy"],
            "tile_weights": [None] * len(times),
            "tile_assignments": [None] * len(times),
        })
        return self._entropy_results
