"""SpatialEntropyAnalyzer: per-frame normalised Shannon entropy of the FoV-weighted tile
histogram, averaged over the configured lattice sizes (reference
analyzers/spatial_entropy.py:107-164), computed by the HIP engine in one call per video."""

from __future__ import annotations

import logging
import time

import pandas as pd

from .. import _native
from ..data_types import ValidationError
from .._results import TileAssignments, TileWeights
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
            if kind == "grid":
                res = self._get_plan().spatial(mu=a, mv=b, want_assign=True, want_weights=True)
            else:
                plan = self._get_plan(dir_table=b)
                try:
                    res = plan.spatial(ids=a, want_assign=True, want_weights=True)
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
        self._entropy_results = pd.DataFrame({
            "time": times,
            "entropy": res["entropy"],
            "tile_weights": [TileWeights(tiles, row) for row in res["weights"]],
            "tile_assignments": [TileAssignments(names, row) for row in res["assign"]],
        })
        return self._entropy_results
