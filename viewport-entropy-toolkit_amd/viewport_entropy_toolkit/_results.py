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
import pandas as pd
from pandas.api.extensions import ExtensionArray, ExtensionDtype, register_extension_dtype


class _RowView(Mapping):
    """Base of the per-frame views: holds references only; everything else is computed on access
    (a 30 000-frame result builds 60 000 of these, so construction must stay trivial)."""

    __slots__ = ("_keys", "_row", "_dict")

    def __init__(self, keys, row):
        self._keys = keys
        self._row = row
        self._dict = None

    def _mask(self):
        raise NotImplementedError

    def _value(self, i):
        raise NotImplementedError

    def _items(self):
        return ((self._keys[int(i)], self._value(i)) for i in np.nonzero(self._mask())[0])

    def _materialise(self) -> dict:
        if self._dict is None:           # built once: dict(view) asks for every key in turn
            self._dict = dict(self._items())
        return self._dict

    def __getitem__(self, key):
        return self._materialise()[key]

    def __iter__(self):
        return iter(self._materialise())

    def __len__(self):
        return int(np.count_nonzero(self._mask())) if self._dict is None else len(self._dict)

    def __repr__(self):
        return repr(self._materialise())


class TileWeights(_RowView):
    """{tile Vector: weight} for the tiles a frame touched (float sums, or int counts)."""

    __slots__ = ("_as_int",)

    def __init__(self, tiles: Sequence, row: np.ndarray, as_int: bool = False):
        self._keys = tiles
        self._row = row
        self._dict = None
        self._as_int = as_int

    def _mask(self):
        # dense convention of the C-ABI (include/vet.h): -0.0 = a key of the reference's dict whose value is 0.0
        # (a tile in some user's FoV whose weight underflowed, entropy_utils.py:131-135), +0.0 = no key
        if self._row.dtype.kind == "f":
            return (self._row > 0) | np.signbit(self._row)
        return self._row > 0

    def _value(self, i):
        return int(self._row[i]) if self._as_int else float(self._row[i]) + 0.0


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


# --------------------------------------------------------------------------------------------
# Device-resident rows and the DataFrame column over them
# --------------------------------------------------------------------------------------------
class DeviceRows:
    """Row ``i`` of a dense ``[T, ...]`` output that lives in device memory: fetched in blocks of ``block``
    rows on first access, the last few blocks kept (a frame-by-frame walk costs one copy per block)."""

    def __init__(self, result, which: int, n_rows: int, block: int = 256, keep: int = 4):
        self._result, self._which, self._n, self._block, self._keep = result, which, int(n_rows), block, keep
        self._cache = {}

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        i = int(i)
        if not 0 <= i < self._n:
            raise IndexError(i)
        b = i // self._block
        blk = self._cache.get(b)
        if blk is None:
            r0 = b * self._block
            blk = self._result.rows(self._which, r0, min(self._block, self._n - r0))
            if len(self._cache) >= self._keep:
                self._cache.pop(next(iter(self._cache)))
            self._cache[b] = blk
        return blk[i - b * self._block]

    def dense(self) -> np.ndarray:
        """The whole output as one host array (one copy)."""
        return self._result.rows(self._which, 0, self._n)


@register_extension_dtype
class FrameDictDtype(ExtensionDtype):
    """dtype of a column whose cells are per-frame result dictionaries (``Mapping`` views)."""
    name = "frame_dict"
    type = Mapping
    kind = "O"
    na_value = None

    @classmethod
    def construct_array_type(cls):
        return FrameDictArray


class FrameDictArray(ExtensionArray):
    """A DataFrame column of per-frame dict views built on access: the column itself is (rows, view factory,
    row index) — no Python object per frame until a cell is read (the reference builds T dicts eagerly,
    analyzers/spatial_entropy.py:158-161; 60 000 view objects cost ~35 ms at BASELINE config 3)."""

    def __init__(self, rows, make, index=None):
        self._rows, self._make = rows, make
        self._index = np.arange(len(rows), dtype=np.int64) if index is None else np.asarray(index, dtype=np.int64)

    # --- pandas interface
    @property
    def dtype(self):
        return FrameDictDtype()

    def __len__(self):
        return len(self._index)

    @property
    def nbytes(self):
        return int(self._index.nbytes)

    def _cell(self, k):
        return None if k < 0 else self._make(self._rows[int(k)])

    def __getitem__(self, item):
        if isinstance(item, (int, np.integer)):
            return self._cell(self._index[item])
        item = pd.api.indexers.check_array_indexer(self, item)
        return FrameDictArray(self._rows, self._make, self._index[item])

    def __iter__(self):
        return (self._cell(k) for k in self._index)

    def isna(self):
        return self._index < 0

    def take(self, indices, allow_fill=False, fill_value=None):
        indices = np.asarray(indices, dtype=np.int64)
        if allow_fill:
            out = np.where(indices < 0, -1, self._index[np.where(indices < 0, 0, indices)] if len(self._index) else -1)
        else:
            out = self._index[indices]
        return FrameDictArray(self._rows, self._make, out)

    def copy(self):
        return FrameDictArray(self._rows, self._make, self._index.copy())

    @classmethod
    def _concat_same_type(cls, to_concat):
        first = to_concat[0]
        if any(a._rows is not first._rows for a in to_concat):
            raise TypeError("cannot concatenate result columns of different runs")
        return cls(first._rows, first._make, np.concatenate([a._index for a in to_concat]))

    @classmethod
    def _from_sequence(cls, scalars, *, dtype=None, copy=False):
        cells = list(scalars)
        return cls(cells, lambda c: c, None)

    def _formatter(self, boxed=False):
        return repr

    def __eq__(self, other):  # element-wise, as pandas expects
        return np.array([a == b for a, b in zip(self, other)], dtype=bool)

    def to_numpy(self, dtype=None, copy=False, na_value=None):
        out = np.empty(len(self), dtype=object)
        for i, v in enumerate(self):
            out[i] = v
        return out

    def __array__(self, dtype=None, copy=None):
        return self.to_numpy()
