"""TransitionEntropyAnalyzer: per-frame entropy of the (t-1 -> t) nearest-tile transitions of
the users present in both frames (reference analyzers/transition_entropy.py:107-175 and
utilities/entropy_utils.py:213-332), computed by the HIP engine in one call per video.
Row 0 of the frame table only seeds the prior, so the result has T-1 rows."""

from __future__ import annotations

import logging
import time

import pandas as pd

from .. import _native
from ..data_types import ValidationError
from .._results import DeviceRows, FrameDictArray, TilePairs, TileWeights
from ._base import _EntropyAnalyzerBase

logger = logging.getLogger(__name__)


class TransitionEntropyAnalyzer(_EntropyAnalyzerBase):
    """Drop-in analyzer with the reference's result schema; ``tile_weights`` holds the user
    count per source tile and ``tile_assignments`` the (prior, current) tile index pairs."""

    _logger = logger

    @staticmethod
    def _empty_row_error(kind, a, b) -> Exception:
        """The exception the reference raises at the FIRST frame pair without a common user: a frame whose dict is
        empty -> ValidationError("Empty vector dictionary") (utilities/entropy_utils.py:239-240); both frames have users
        but nobody is in both -> the division by the zero total weight (:322-327)."""
        import numpy as np
        present = (a >= 0) if kind != "grid" else ~(np.isnan(a) | np.isnan(b))
        common = (present[:-1] & present[1:]).any(axis=1)
        r = int(np.argmin(common))                 # first row without a common user
        if not present[r].any() or not present[r + 1].any():
            return ValidationError("Empty vector dictionary")
        return ZeroDivisionError("division by zero")          # `1 / total_weight` with the int 0 (:326)

    def compute_entropy(self) -> pd.DataFrame:
        kind, times, a, b, names = self._samples()
        t_start = time.perf_counter()
        try:
            if kind == "grid":
                res = self._get_plan().transition_resident(mu=a, mv=b)
            else:
                plan = self._get_plan(dir_table=b)
                try:
                    res = plan.transition_resident(ids=a)
                finally:
                    plan.close()
        except _native.NativeError as e:
            if e.code == _native.VET_ERR_RANGE:
                raise ValidationError(str(e))
            if e.code == _native.VET_ERR_EMPTY:
                raise self._empty_row_error(kind, a, b)
            raise
        self._record_compute(time.perf_counter() - t_start, a.size, len(res["entropy"]))
        tiles = self._fibonacci_vectors[self.config.tile_counts[0]]
        R = len(res["entropy"])
        self._device_result = res["result"]
        self._entropy_results = pd.DataFrame({
            "time": times[1:],
            "entropy": res["entropy"],
            "tile_weights": FrameDictArray(DeviceRows(res["result"], 1, R), lambda row: TileWeights(tiles, row, as_int=True)),
            "tile_assignments": FrameDictArray(DeviceRows(res["result"], 0, R), lambda row: TilePairs(names, row)),
        })
        return self._entropy_results
