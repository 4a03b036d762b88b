"""Lazy per-frame result dictionaries over the engine's dense outputs.

The reference returns, for every frame, a ``dict`` tile ``Vector`` -> weight and a ``dict``
user -> tile index (analyzers/spatial_entropy.py:158-163).  Building T Python dicts costs far
more than the kernels, so the DataFrame cells are read-only ``Mapping`` views over the dense
``[T, n0]`` / ``[T, U]`` arrays; they compare equal to the dicts the reference builds.
"""

from __future__ import annotations

from collections.abc import Mapping
from typing import List, Sequence

import numpy as np


class _RowView(Mapping):
    """Base of the per-frame views: holds references only; everything else is computed on access
    (a 30 000-frame result builds 60 000 of these, so construction must stay trivial)."""

    __slots__ = ("_keys", "_row")

    def __init__(self, keys, row):
        self._keys = keys
        self._row = row

    def _mask(self):
        raise NotImplementedError

    def _value(self, i):
        raise NotImplementedError

    def _items(self):
        return ((self._keys[int(i)], self._value(i)) for i in np.nonzero(self._mask())[0])

    def _materialise(self) -> dict:
        return dict(self._items())

    def __getitem__(self, key):
        return self._materialise()[key]

    def __iter__(self):
        return (k for k, _ in self._items())

    def __len__(self):
        return int(np.count_nonzero(self._mask()))

    def __repr__(self):
        return repr(self._materialise())


class TileWeights(_RowView):
    """{tile Vector: weight} for the tiles a frame touched (float sums, or int counts)."""

    __slots__ = ("_as_int",)

    def __init__(self, tiles: Sequence, row: np.ndarray, as_int: bool = False):
        self._keys = tiles
        self._row = row
        self._as_int = as_int

    def _mask(self):
        return self._row > 0

    def _value(self, i):
        return int(self._row[i]) if self._as_int else float(self._row[i])


class TileAssignments(_RowView):
    """{user identifier: nearest tile index} for the users present in a frame."""

    __slots__ = ()

    def _mask(self):
        return self._row >= 0

    def _value(self, i):
        return int(self._row[i])


class TilePairs(_RowView):
    """{user identifier: (prior tile index, current tile index)} for users in both frames."""

    __slots__ = ()

    def _mask(self):
        return self._row[:, 0] >= 0

    def _value(self, i):
        return (int(self._row[i, 0]), int(self._row[i, 1]))
