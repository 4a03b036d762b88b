"""SpatialEntropyAnalyzer: per-frame normalised Shannon entropy of the FoV-weighted tile
histogram, averaged over the configured lattice sizes (reference
analyzers/spatial_entropy.py:107-164), computed by the HIP engine in one call per video."""

from __future__ import annotations

import logging
import time

import pandas as pd

from .. import _native
from ..data_types import ValidationError
from .._results import DeviceRows, FrameDictArray, TileAssignments, TileWeights
from ._base import _EntropyAnalyzerBase

logger = logging.getLogger(__name__)


class SpatialEntropyAnalyzer(_EntropyAnalyzerBase):
    """Drop-in analyzer: ``process_directory`` / ``compute_entropy`` / ``create_visualization`` /
    ``run_analysis`` with the reference's result schema
    (``time``, ``entropy``, ``tile_weights``, ``tile_assignments``)."""

    _logger = logger

    def compute_entropy(self) -> pd.DataFrame:
        kind, times, a, b, names = self._samples()
        t_start = time.perf_counter()
        try:
            # only the entropy series crosses PCIe; tile weights / assignments stay in device memory and are
            # fetched by frame when a cell of the result is read
            if kind == "grid":
                res = self._get_plan().spatial_resident(mu=a, mv=b)
            else:
                plan = self._get_plan(dir_table=b)
                try:
                    res = plan.spatial_resident(ids=a)
                finally:
                    plan.close()
        except _native.NativeError as e:
            if e.code == _native.VET_ERR_RANGE:
                raise ValidationError(str(e))
            if e.code == _native.VET_ERR_EMPTY:
                raise ValidationError("Empty vector dictionary")
            raise
        self._record_compute(time.perf_counter() - t_start, a.size, len(res["entropy"]))
        tiles = self._fibonacci_vectors[self.config.tile_counts[0]]
        T = len(res["entropy"])
        self._device_result = res["result"]
        self._entropy_results = pd.DataFrame({
            "time": times,
            "entropy": res["entropy"],
            "tile_weights": FrameDictArray(DeviceRows(res["result"], 1, T), lambda row: TileWeights(tiles, row)),
            "tile_assignments": FrameDictArray(DeviceRows(res["result"], 0, T), lambda row: TileAssignments(names, row)),
        })
        return self._entropy_results
