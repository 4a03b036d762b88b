"""ctypes wrapper of oracle/vet_oracle.c (TEST INFRASTRUCTURE; see that file's header)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

from . import vet_oracle as vo

HERE = Path(__file__).resolve().parent
LIB = Path(os.environ.get("VET_ORACLE_LIB", HERE / "_build" / "libvet_oracle.so"))


def set_threads(n: int) -> None:
    """OpenMP threads of the spatial port (frames are spread over them)."""
    lib = load()
    try:
        omp = C.CDLL("libgomp.so.1")
        omp.omp_set_num_threads(int(n))
    except OSError:
        pass
    del lib


def load(build: bool = True):
    if not LIB.exists() and build:
        # several ranks of one job may get here together (bench.py's parity gate at N > 1): one of them builds
        import fcntl
        (HERE / "_build").mkdir(exist_ok=True)
        with open(HERE / "_build" / ".lock", "w") as lock:
            fcntl.flock(lock, fcntl.LOCK_EX)
            try:
                if not LIB.exists():
                    subprocess.run(["make", "-C", str(HERE)], check=True, capture_output=True)
            finally:
                fcntl.flock(lock, fcntl.LOCK_UN)
    lib = C.CDLL(str(LIB))
    lib.oracle_spatial.restype = C.c_int
    lib.oracle_transition.restype = C.c_int
    return lib


def _common(mu, mv, W, H, tile_counts):
    mu = np.ascontiguousarray(mu, dtype=np.float64)
    mv = np.ascontiguousarray(mv, dtype=np.float64)
    grid = np.ascontiguousarray(vo.direction_grid(W, H).reshape(-1, 3))
    tiles = [np.ascontiguousarray(vo.fibonacci_lattice(tc)) for tc in tile_counts]
    n = np.asarray([len(t) for t in tiles], dtype=np.int32)
    ptrs = (C.c_void_p * len(tiles))(*[t.ctypes.data for t in tiles])
    return mu, mv, grid, tiles, n, ptrs


def spatial_series(mu, mv, W, H, tile_counts, fov_angle=120.0, power_factor=2.0,
                   use_weight_distribution=True, want_weights=False):
    lib = load()
    mu, mv, grid, tiles, n, ptrs = _common(mu, mv, W, H, tile_counts)
    T, U = mu.shape
    ent = np.empty(T)
    assign = np.empty((T, U), dtype=np.int32)
    weights = np.empty((T, n[0])) if want_weights else None
    rc = lib.oracle_spatial(C.c_void_p(mu.ctypes.data), C.c_void_p(mv.ctypes.data), T, U, W, H,
                            C.c_void_p(grid.ctypes.data), len(tiles), C.c_void_p(n.ctypes.data), ptrs,
                            C.c_double(fov_angle), C.c_double(power_factor), int(use_weight_distribution),
                            C.c_void_p(ent.ctypes.data), C.c_void_p(assign.ctypes.data),
                            C.c_void_p(weights.ctypes.data) if want_weights else None)
    if rc:
        raise ValueError(f"oracle_spatial rc={rc}")
    return ent, assign, weights


def transition_series(mu, mv, W, H, tile_counts):
    lib = load()
    mu, mv, grid, tiles, n, ptrs = _common(mu, mv, W, H, tile_counts)
    T, U = mu.shape
    ent = np.empty(T - 1)
    pairs = np.empty((T - 1, U, 2), dtype=np.int32)
    rc = lib.oracle_transition(C.c_void_p(mu.ctypes.data), C.c_void_p(mv.ctypes.data), T, U, W, H,
                               C.c_void_p(grid.ctypes.data), len(tiles), C.c_void_p(n.ctypes.data), ptrs,
                               C.c_void_p(ent.ctypes.data), C.c_void_p(pairs.ctypes.data))
    if rc == -4:
        raise ZeroDivisionError("float division by zero")
    if rc:
        raise ValueError(f"oracle_transition rc={rc}")
    return ent, pairs
