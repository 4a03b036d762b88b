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
    __slots__ = ("_keys", "_row", "_present")

    def _items(self):
        raise NotImplementedError

    def _materialise(self) -> dict:
        return dict(self._items())

    def __getitem__(self, key):
        return self._materialise()[key]

    def __iter__(self):
        return (k for k, _ in self._items())

    def __len__(self):
        return int(np.count_nonzero(self._present))

    def __repr__(self):
        return repr(self._materialise())


class TileWeights(_RowView):
    """{tile Vector: weight} for the tiles a frame touched."""

    def __init__(self, tiles: Sequence, row: np.ndarray, as_int: bool = False):
        self._keys, self._row, self._present = tiles, row, row > 0
        self._as_int = as_int

    __slots__ = ("_as_int",)

    def _items(self):
        cast = int if self._as_int else float
        return ((self._keys[int(i)], cast(self._row[i])) for i in np.nonzero(self._present)[0])


class TileAssignments(_RowView):
    """{user identifier: nearest tile index} for the users present in a frame."""

    def __init__(self, users: List[str], row: np.ndarray):
        self._keys, self._row, self._present = users, row, row >= 0

    def _items(self):
        return ((self._keys[int(i)], int(self._row[i])) for i in np.nonzero(self._present)[0])


class TilePairs(_RowView):
    """{user identifier: (prior tile index, current tile index)} for users in both frames."""

    def __init__(self, users: List[str], row: np.ndarray):
        self._keys, self._row, self._present = users, row, row[:, 0] >= 0

    def _items(self):
        return ((self._keys[int(i)], (int(self._row[i, 0]), int(self._row[i, 1])))
                for i in np.nonzero(self._present)[0])
