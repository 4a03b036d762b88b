"""Dense, vectorised ingest: per-user CSV tracks -> frame-major ``mu[T][U]`` / ``mv[T][U]``.

Same semantics as the reference's ``process_viewport_data`` + ``format_trajectory_data``
(utilities/data_utils.py:289-410) without its per-row Python objects and O(T^2) list search:

* rows with a NaN in time / 2dmu / 2dmv are dropped, time is shifted to start at 0;
* any 2dmu / 2dmv outside [0, 1] fails the whole file; W and H must be even;
* frame time = ``round(time, 1)``; the frame index is the FIRST-APPEARANCE order of the
  rounded times while walking users in the given order (not sorted);
* several rows of one user in the same frame: the last row wins;
* a (frame, user) cell without a row is absent (NaN here, ``None`` in the reference).

The arrays produced here are exactly what the C-ABI consumes; the quantisation to pixels,
directions and tiles happens on the device.
"""

from __future__ import annotations

from pathlib import Path
import logging
import os
from typing import List, Sequence, Tuple, Union

import numpy as np
import pandas as pd

from .data_types import ValidationError


def check_video_dimensions(width: int, height: int) -> None:
    if width <= 0 or height <= 0:
        raise ValidationError("Video dimensions must be positive")
    if width % 2 != 0 or height % 2 != 0:
        raise ValidationError("Video dimensions must be even numbers")


def check_normalized(normalized: np.ndarray, dimension: int) -> None:
    """The two checks of normalize_to_pixel (data_utils.py:256-259) without the conversion; NaN passes (it compares
    False on both sides, as in the reference)."""
    normalized = np.asarray(normalized)
    with np.errstate(invalid="ignore"):
        if np.any((normalized < 0) | (normalized > 1)):
            raise ValidationError("Normalized coordinates must be between 0 and 1")
    if dimension <= 0:
        raise ValidationError("Dimension must be positive")


def to_pixels(normalized: np.ndarray, dimension: int) -> np.ndarray:
    """(v * dimension) truncated to int; v must lie in [0, 1]."""
    normalized = np.asarray(normalized)
    check_normalized(normalized, dimension)
    return (normalized * dimension).astype(int)


def _clean(time, mu, mv, labels, filepath: Path, width: int, height: int):
    """dropna, time shift, range and dimension checks of process_viewport_data (data_utils.py:318-331)."""
    keep = ~(np.isnan(time) | np.isnan(mu) | np.isnan(mv))
    if not keep.all():
        labels, time, mu, mv = labels[keep], time[keep], mu[keep], mv[keep]
    if len(time) == 0:
        raise ValidationError(f"No valid data found in {filepath}")
    time = time - time.min()
    to_pixels(mu, width)                # range / dimension checks, in the reference's order
    to_pixels(mv, height)
    check_video_dimensions(width, height)
    return labels, time, mu, mv, filepath.stem


def _read_columns_pandas(filepath: Path):
    data = pd.read_csv(filepath, usecols=["time", "2dmu", "2dmv"])
    return (data["time"].to_numpy(dtype=np.float64), data["2dmu"].to_numpy(dtype=np.float64),
            data["2dmv"].to_numpy(dtype=np.float64), data.index.to_numpy())


def read_samples(filepath: Union[str, Path], width: int, height: int, columns=None):
    """One user's CSV -> ``(row labels, time, mu, mv, identifier)`` float64 arrays, cleaned and
    validated as ``process_viewport_data`` does (data_utils.py:289-342: NaN rows dropped, time
    shifted to start at 0, coordinates checked against [0, 1], dimensions checked) — without
    building the per-row DataFrame columns the engine never reads.  ``columns`` = raw
    ``(time, mu, mv)`` already parsed by the native loader."""
    try:
        filepath = Path(filepath)
        if columns is None:
            if not filepath.exists():
                raise FileNotFoundError(f"File not found: {filepath}")
            time, mu, mv, labels = _read_columns_pandas(filepath)
        else:
            time, mu, mv = columns
            labels = np.arange(len(time))
        return _clean(time, mu, mv, labels, filepath, width, height)
    except Exception as e:  # noqa: BLE001 - the reference funnels every failure into ValidationError
        raise ValidationError(f"Error processing viewport data: {str(e)}")


_native_state = {"verified": None}      # None = not checked yet, True / False = result of the check against pandas


def read_directory(files: Sequence[Path], width: int, height: int, threads: int = 0):
    """All user files of one video -> list of ``read_samples`` results, in ``files`` order.

    ``VET_CSV_PARSER`` = ``native`` (default when libvet_hip.so is built): the C-ABI's threaded
    loader (vet_csv_read_tracks) parses the plain numeric files, pandas the ones it declines, and
    the first file a process reads is parsed both ways and compared, so a pandas whose converter
    differs sends everything through pandas; ``pandas``: ``pd.read_csv`` for every file, as the reference."""
    files = [Path(f) for f in files]
    mode = os.environ.get("VET_CSV_PARSER", "native")
    parsed = None
    if mode != "pandas" and files and _native_state["verified"] is not False:
        try:
            from . import _native
            if threads <= 0:      # the CPUs this process may run on, not the machine's
                threads = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1), 32)
            parsed = _native.read_tracks(files, threads)
        except Exception:  # noqa: BLE001 - library not built: the reference's own parser does the job
            parsed = None
        if parsed is not None and _native_state["verified"] is None:
            # once per process: the converter must agree with this pandas bit for bit on a real file
            first = next((i for i, p in enumerate(parsed) if p[0] == 0), None)
            if first is not None:
                try:
                    t, a, b, _ = _read_columns_pandas(files[first])
                    same = all(np.array_equal(x, y, equal_nan=True) for x, y in zip((t, a, b), parsed[first][1:]))
                except Exception:  # noqa: BLE001
                    same = False
                _native_state["verified"] = same
                if not same:
                    logging.getLogger(__name__).warning("native CSV loader disagrees with pandas on %s; using pandas",
                                                        files[first])
        if _native_state["verified"] is False:
            parsed = None
    out = []
    for i, fp in enumerate(files):
        cols = parsed[i][1:] if parsed is not None and parsed[i][0] == 0 else None
        out.append(read_samples(fp, width, height, columns=cols))
    return out


def frame_of_samples(labels, time, mu, mv, width: int, height: int) -> pd.DataFrame:
    """The reference's per-user frame (time, 2dmu, 2dmv, pixel_x, pixel_y, lon, lat) from cleaned samples."""
    px, py = to_pixels(mu, width), to_pixels(mv, height)
    return pd.DataFrame({"time": time, "2dmu": mu, "2dmv": mv, "pixel_x": px, "pixel_y": py,
                         "lon": (px / width) * 360 - 180, "lat": 90 - (py / height) * 180},
                        index=pd.Index(labels))


def read_track(filepath: Union[str, Path], width: int, height: int) -> Tuple[pd.DataFrame, str]:
    """One user's CSV -> cleaned DataFrame (time, 2dmu, 2dmv, pixel_x, pixel_y, lon, lat)."""
    labels, time, mu, mv, identifier = read_samples(filepath, width, height)
    try:
        return frame_of_samples(labels, time, mu, mv, width, height), identifier
    except Exception as e:  # noqa: BLE001
        raise ValidationError(f"Error processing viewport data: {str(e)}")


def frame_keys(time: np.ndarray) -> np.ndarray:
    """Integer identity of ``round(time, 1)`` (= rint(time*10); the float is key/10)."""
    return np.rint(np.asarray(time, dtype=np.float64) * 10.0).astype(np.int64)


def build_dense(tracks: Sequence[Tuple[np.ndarray, np.ndarray, np.ndarray]]):
    """tracks: per user (time, mu, mv) already cleaned.  Returns (frame_times[T], mu[T,U], mv[T,U])."""
    if not tracks:
        raise ValidationError("No trajectory data provided")
    known = np.empty(0, dtype=np.int64)           # frame keys in first-appearance order
    known_sorted = known
    per_user = []
    for time, _, _ in tracks:
        keys = frame_keys(time)
        increasing = len(keys) < 2 or bool(np.all(keys[1:] > keys[:-1]))
        if increasing:                            # common case: one row per frame, in time order
            uniq = keys
            same = len(uniq) == len(known_sorted) and bool(np.array_equal(uniq, known_sorted))
        else:
            uniq, first = np.unique(keys, return_index=True)
            same = len(uniq) == len(known_sorted) and bool(np.array_equal(uniq, known_sorted))
            if not same:
                uniq = uniq[np.argsort(first, kind="stable")]       # this user's first-appearance order
        if not same:
            if len(known):
                uniq = uniq[~np.isin(uniq, known_sorted, assume_unique=True)]
            if len(uniq):
                known = np.concatenate([known, uniq])
                known_sorted = np.sort(known)
        per_user.append((keys, increasing))
    T, U = len(known), len(tracks)
    order = np.argsort(known, kind="stable")
    sorted_keys = known[order]
    # filled user-major (contiguous writes per track), transposed to frame-major once at the end
    mu = np.full((U, T), np.nan)
    mv = np.full((U, T), np.nan)
    for u, ((_, a, b), (keys, increasing)) in enumerate(zip(tracks, per_user)):
        frame = order[np.searchsorted(sorted_keys, keys)]
        a = np.asarray(a, dtype=np.float64)
        b = np.asarray(b, dtype=np.float64)
        if increasing:                            # no duplicate frames for this user
            mu[u, frame] = a
            mv[u, frame] = b
            continue
        # last row of a frame wins: walk reversed, keep first occurrence
        _, idx = np.unique(frame[::-1], return_index=True)
        rows = len(frame) - 1 - idx
        mu[u, frame[rows]] = a[rows]
        mv[u, frame[rows]] = b[rows]
    mu = np.ascontiguousarray(mu.T)
    mv = np.ascontiguousarray(mv.T)
    return known.astype(np.float64) / 10.0, mu, mv


def tracks_from_frames(trajectory_data: List[Tuple[str, pd.DataFrame]]):
    return [(d["time"].to_numpy(dtype=np.float64), d["2dmu"].to_numpy(dtype=np.float64),
             d["2dmv"].to_numpy(dtype=np.float64)) for _, d in trajectory_data]
