"""ctypes binding of the HIP engine's C-ABI (include/vet.h, libvet_hip.so).

This module is the only place the package touches native code.  It has no CPU fallback: if
the shared library has not been built, or no gfx950 device is visible, every compute entry
point raises ``NativeUnavailable`` with the reason.
"""

from __future__ import annotations

import ctypes as C
import os
import threading
from pathlib import Path
from typing import Optional, Sequence

import numpy as np

from . import _quantiser

_PKG_DIR = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get("VET_HIP_LIBRARY", _PKG_DIR.parent / "lib" / "libvet_hip.so"))

VET_OK, VET_ERR_INVALID, VET_ERR_DEVICE, VET_ERR_RANGE, VET_ERR_EMPTY, VET_ERR_UNSUPPORTED = 0, -1, -2, -3, -4, -5
KERNEL_IDS = {"k_grid_dirs": 0, "k_nearest_lut": 1, "k_spatial": 2, "k_transition": 3, "k_finalize": 4,
              "k_wtab": 5, "k_weights": 6}


class NativeUnavailable(RuntimeError):
    """The HIP extension is missing or no MI355X is visible."""


class NativeError(RuntimeError):
    """A C-ABI call failed; ``code`` is the VET_ERR_* value."""

    def __init__(self, code: int, message: str):
        super().__init__(message)
        self.code = code


class Video(C.Structure):
    """include/vet.h: vet_video (device pointers of one video of a batch)."""
    _fields_ = [("d_mu", C.c_void_p), ("d_mv", C.c_void_p), ("n_users", C.c_int), ("n_frames", C.c_int),
                ("d_entropy", C.c_void_p), ("d_assign", C.c_void_p), ("d_present", C.c_void_p)]


class Track(C.Structure):
    """include/vet.h: vet_track (one parsed CSV file, host memory owned by the library)."""
    _fields_ = [("time", C.POINTER(C.c_double)), ("mu", C.POINTER(C.c_double)), ("mv", C.POINTER(C.c_double)),
                ("n_rows", C.c_int64), ("status", C.c_int)]


VET_CSV_OK, VET_CSV_FALLBACK, VET_CSV_IO = 0, 1, 2


class _PlanDesc(C.Structure):
    _fields_ = [
        ("video_width", C.c_int), ("video_height", C.c_int),
        ("h_lon_cos", C.c_void_p), ("h_lon_sin", C.c_void_p),
        ("h_lat_sin", C.c_void_p), ("h_lat_cos", C.c_void_p),
        ("h_dir_table", C.c_void_p), ("n_dirs", C.c_int64),
        ("n_lattices", C.c_int), ("n_tiles", C.c_void_p),
        ("h_tiles", C.c_void_p), ("h_max_entropy", C.c_void_p),
        ("fov_angle", C.c_double), ("max_angular_distance", C.c_double),
        ("power_factor", C.c_double), ("use_weight_distribution", C.c_int),
        ("h_bin_lut", C.c_void_p), ("n_norm_tiles", C.c_void_p),
    ]


# name -> (restype, argtypes); also the list tests check against include/vet.h
_P, _I, _I64, _D, _SZ = C.c_void_p, C.c_int, C.c_int64, C.c_double, C.c_size_t
SIGNATURES = {
    "vet_version": (_I, []),
    "vet_last_error": (C.c_char_p, []),
    "vet_device_count": (_I, []),
    "vet_create": (_I, [_I, C.POINTER(_P)]),
    "vet_destroy": (_I, [_P]),
    "vet_synchronize": (_I, [_P]),
    "vet_device_pci_bus_id": (_I, [_P, C.c_char_p, _I]),
    "vet_profile_enable": (_I, [_P, _I]),
    "vet_profile_reset": (_I, [_P]),
    "vet_profile_get": (_I, [_P, _I, C.POINTER(_D), C.POINTER(_I64)]),
    "vet_kernel_name": (C.c_char_p, [_I]),
    "vet_malloc": (_I, [_P, _SZ, C.POINTER(_P)]),
    "vet_free": (_I, [_P, _P]),
    "vet_memcpy_h2d": (_I, [_P, _P, _P, _SZ]),
    "vet_memcpy_d2h": (_I, [_P, _P, _P, _SZ]),
    "vet_plan_create": (_I, [_P, C.POINTER(_PlanDesc), C.POINTER(_P)]),
    "vet_plan_destroy": (_I, [_P]),
    "vet_plan_n_dirs": (_I64, [_P]),
    "vet_plan_set_table_policy": (_I, [_P, _I]),
    "vet_plan_set_raw_weights": (_I, [_P, _I]),
    "vet_plan_table_stride": (_I, [_P, _I]),
    "vet_plan_table_rows": (_I64, [_P]),
    "vet_plan_last_formulation": (_I, [_P, _I]),
    "vet_plan_error_bounds": (_I, [_P, _I, C.POINTER(_D), C.POINTER(_D)]),
    "vet_plan_read_dirs": (_I, [_P, _P]),
    "vet_plan_read_nearest": (_I, [_P, _I, _P]),
    "vet_spatial_entropy": (_I, [_P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P]),
    "vet_spatial_entropy_ids": (_I, [_P, _P, _I, _I, _P, _P, _P, _P, _P, _P]),
    "vet_transition_entropy": (_I, [_P, _P, _P, _I, _I, _P, _P, _P, _P, _P, _P]),
    "vet_transition_entropy_ids": (_I, [_P, _P, _I, _I, _P, _P, _P, _P, _P, _P]),
    "vet_spatial_entropy_batch": (_I, [_P, _I, _P, _P, _P]),
    "vet_spatial_entropy_batch_host": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P]),
    "vet_transition_entropy_batch": (_I, [_P, _I, _P, _P, _P]),
    "vet_transition_entropy_batch_host": (_I, [_P, _I, _P, _P, _P, _P, _P, _P, _P]),
    "vet_spatial_entropy_host": (_I, [_P, _P, _P, _P, _I, _I, _P, _P, _P, _P]),
    "vet_transition_entropy_host": (_I, [_P, _P, _P, _P, _I, _I, _P, _P, _P, _P]),
    "vet_spatial_entropy_host_resident": (_I, [_P, _P, _P, _P, _I, _I, _P, _P, C.POINTER(_P)]),
    "vet_transition_entropy_host_resident": (_I, [_P, _P, _P, _P, _I, _I, _P, _P, C.POINTER(_P)]),
    "vet_result_fetch": (_I, [_P, _I, _I64, _I64, _P]),
    "vet_result_free": (_I, [_P]),
    "vet_fb_tile_boundaries": (_I, [_P, _P, _I, _I, _P, _P]),
    "vet_angular_distances": (_I, [_P, _P, _I64, _P, _I, _P]),
    "vet_csv_read_tracks": (_I, [_I, C.POINTER(C.c_char_p), C.POINTER(Track), _I]),
    "vet_csv_free_tracks": (None, [_I, C.POINTER(Track)]),
}

_lib = None
_lib_lock = threading.Lock()


def _preload_hip_runtime() -> Optional[str]:
    """One HIP runtime per process.  libvet_hip.so needs ``libamdhip64.so.7``; a PyTorch-ROCm wheel ships its own
    copy (``torch/lib/libamdhip64.so``, same SONAME) and a process that ends up with both — this library first,
    ``torch.cuda`` later — fails in torch's ``_cuda_init`` with "No HIP GPUs are available".  So when torch is
    installed its runtime is loaded first, globally, and the dynamic linker binds libvet_hip.so to it by SONAME
    (torch is NOT imported; if it already is, this is the runtime it loaded).  ``VET_HIP_RUNTIME=system`` keeps the
    system runtime.  Returns the path that was preloaded, or None."""
    if os.environ.get("VET_HIP_RUNTIME", "").lower() == "system":
        return None
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return None
    cand = Path(spec.origin).parent / "lib" / "libamdhip64.so"
    if not cand.exists():
        return None
    try:
        C.CDLL(str(cand), mode=C.RTLD_GLOBAL)
    except OSError:
        return None
    return str(cand)


HIP_RUNTIME_PRELOADED: Optional[str] = None


def load_library():
    """dlopen libvet_hip.so and declare every prototype; raises NativeUnavailable."""
    global _lib, HIP_RUNTIME_PRELOADED
    with _lib_lock:
        if _lib is not None:
            return _lib
        if not LIB_PATH.exists():
            raise NativeUnavailable(
                f"HIP extension not found at {LIB_PATH}. Build it with "
                f"`make -C {_PKG_DIR.parent / 'csrc'}` (or __graft_entry__.build()); "
                "this package has no CPU compute path.")
        HIP_RUNTIME_PRELOADED = _preload_hip_runtime()
        try:
            lib = C.CDLL(str(LIB_PATH))
        except OSError as e:  # missing ROCm runtime etc.
            raise NativeUnavailable(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        _lib = lib
        return lib


def read_tracks(paths: Sequence, n_threads: int = 0):
    """vet_csv_read_tracks over ``paths``: list of ``(status, time, mu, mv)`` with one FP64 entry per
    data row (NaN = missing); arrays are ``None`` unless status is VET_CSV_OK.  Host code only."""
    lib = load_library()
    n = len(paths)
    if n == 0:
        return []
    c_paths = (C.c_char_p * n)(*[os.fsencode(str(p)) for p in paths])
    tracks = (Track * n)()
    rc = lib.vet_csv_read_tracks(n, c_paths, tracks, int(n_threads))
    if rc != VET_OK:
        raise NativeError(rc, "vet_csv_read_tracks failed")
    try:
        out = []
        for t in tracks:
            if t.status == VET_CSV_OK:
                m = int(t.n_rows)
                cols = [np.frombuffer(C.string_at(ptr, m * 8), dtype=np.float64).copy() if m else np.empty(0)
                        for ptr in (t.time, t.mu, t.mv)]
                out.append((VET_CSV_OK, *cols))
            else:
                out.append((int(t.status), None, None, None))
        return out
    finally:
        lib.vet_csv_free_tracks(n, tracks)


def _check(lib, rc: int):
    if rc != VET_OK:
        msg = lib.vet_last_error()
        raise NativeError(rc, (msg or b"").decode("utf-8", "replace") or f"vet error {rc}")


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


VET_STREAM_LEGACY = 1
TABLE_SAMPLES_PER_DIRECTION = 2      # include/vet.h: VET_TABLE_SAMPLES_PER_DIRECTION (policy 0: table iff samples >= this x directions)


def _stream(handle):
    """hipStream_t handle -> the C-ABI's ``stream`` argument: ``None`` = the engine's own stream;
    0 (torch's default stream) = the legacy null stream; anything else is the stream itself."""
    if handle is None:
        return None
    return C.c_void_p(VET_STREAM_LEGACY if int(handle) == 0 else int(handle))


class Engine:
    """One device context (stream + scratch).  ``Engine.default()`` is per process."""

    _default = None
    _default_lock = threading.Lock()

    def __init__(self, device_id: int = 0):
        self.lib = load_library()
        h = C.c_void_p()
        rc = self.lib.vet_create(device_id, C.byref(h))
        if rc == VET_ERR_DEVICE:
            raise NativeUnavailable((self.lib.vet_last_error() or b"").decode())
        _check(self.lib, rc)
        self.handle = h
        self.device_id = device_id

    @staticmethod
    def default_device_id(n_devices: int) -> int:
        """Device of the per-process default engine: ``VET_DEVICE``, else ``LOCAL_RANK`` (one process per GPU), else 0.
        A value that names no visible device RAISES: falling back to device 0 would silently stack the ranks of a
        multi-GPU job on one GPU."""
        for var in ("VET_DEVICE", "LOCAL_RANK"):
            raw = os.environ.get(var)
            if raw is None:
                continue
            try:
                dev = int(raw)
            except ValueError:
                raise NativeUnavailable(f"{var}={raw!r} is not a device index") from None
            if not 0 <= dev < n_devices:
                raise NativeUnavailable(
                    f"{var}={dev} names no visible device ({n_devices} visible): refusing to fall back to device 0 "
                    "(the ranks of a multi-GPU job would share one GPU); fix the launcher's device visibility or set VET_DEVICE")
            return dev
        return 0

    @classmethod
    def default(cls) -> "Engine":
        with cls._default_lock:
            if cls._default is None:
                n = load_library().vet_device_count()
                if n <= 0:
                    raise NativeUnavailable("no HIP device available; this package has no CPU compute path")
                cls._default = cls(cls.default_device_id(n))
            return cls._default

    def pci_bus_id(self) -> str:
        """PCI bus id of the device this context computes on (include/vet.h: vet_device_pci_bus_id)."""
        buf = C.create_string_buffer(64)
        _check(self.lib, self.lib.vet_device_pci_bus_id(self.handle, buf, 64))
        return buf.value.decode()

    def close(self):
        if getattr(self, "handle", None):
            self.lib.vet_destroy(self.handle)
            self.handle = None

    def synchronize(self):
        _check(self.lib, self.lib.vet_synchronize(self.handle))

    def fb_tile_boundaries(self, tiles: np.ndarray, max_edges: int = 16):
        """k_fb_boundaries: (edges [n, max_edges, 2, 3] NaN padded, count [n]) for lattice Vectors ``tiles`` [n, 3]."""
        tiles = np.ascontiguousarray(tiles, dtype=np.float64).reshape(-1, 3)
        n = len(tiles)
        edges = np.empty((n, max_edges, 2, 3), dtype=np.float64)
        count = np.empty(n, dtype=np.int32)
        _check(self.lib, self.lib.vet_fb_tile_boundaries(self.handle, _ptr(tiles), n, max_edges, _ptr(edges), _ptr(count)))
        return edges, count

    # --- profiling -------------------------------------------------------
    def angular_distances(self, vectors: np.ndarray, tiles: np.ndarray) -> np.ndarray:
        """[m, n] arccos(clip(dot(v/|v|, t/|t|))) — vector_angle_distance of the reference for every pair."""
        vectors = np.ascontiguousarray(vectors, dtype=np.float64).reshape(-1, 3)
        tiles = np.ascontiguousarray(tiles, dtype=np.float64).reshape(-1, 3)
        out = np.empty((len(vectors), len(tiles)), dtype=np.float64)
        if out.size:
            _check(self.lib, self.lib.vet_angular_distances(self.handle, _ptr(vectors), len(vectors), _ptr(tiles), len(tiles), _ptr(out)))
        return out

    def profile_enable(self, on: bool = True):
        _check(self.lib, self.lib.vet_profile_enable(self.handle, int(on)))

    def profile_reset(self):
        _check(self.lib, self.lib.vet_profile_reset(self.handle))

    def profile_get(self, kernel: str):
        ms, n = C.c_double(), C.c_int64()
        _check(self.lib, self.lib.vet_profile_get(self.handle, KERNEL_IDS[kernel], C.byref(ms), C.byref(n)))
        return ms.value, n.value


class DeviceResult:
    """Optional outputs of one run, resident in device memory (include/vet.h: vet_result).  ``rows(which, r0, n)``
    copies rows [r0, r0+n) of output ``which`` (0 = assignments / pairs, 1 = weights / source counts)."""

    def __init__(self, lib, handle, n_rows, shapes, dtypes, engine=None):
        self.lib, self.handle, self.n_rows = lib, handle, int(n_rows)
        self.shapes, self.dtypes = shapes, dtypes
        self.engine = engine           # keeps the device context alive as long as the rows can be fetched

    def rows(self, which: int, row0: int, n: int) -> np.ndarray:
        if self.handle is None:
            raise RuntimeError("the device-resident result has been released")
        out = np.empty((n,) + tuple(self.shapes[which]), dtype=self.dtypes[which])
        _check(self.lib, self.lib.vet_result_fetch(self.handle, which, row0, n, _ptr(out)))
        return out

    def close(self):
        if getattr(self, "handle", None):
            self.lib.vet_result_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


class Plan:
    """Device tables of one analyzer configuration (quantiser, lattices, nearest LUTs)."""

    def __init__(self, engine: Engine, tile_xyz: Sequence[np.ndarray], fov_angle: float, power_factor: float,
                 use_weight_distribution: bool, video_width: int = 0, video_height: int = 0,
                 dir_table: Optional[np.ndarray] = None, bin_luts: Optional[Sequence] = None,
                 bin_counts: Optional[Sequence[int]] = None, bin_max_entropy: Optional[Sequence[float]] = None,
                 bin_norm_tiles: Optional[Sequence[int]] = None):
        """``tile_xyz``: one [n,3] array per lattice.  A *binned* lattice k (naive lat/lon tiling)
        passes ``tile_xyz[k] = None`` with ``bin_luts[k]`` (uint16 [n_dirs] direction -> bin),
        ``bin_counts[k]`` bins and the normaliser ``bin_max_entropy[k]``."""
        self.engine = engine
        self.lib = engine.lib
        bin_luts = list(bin_luts) if bin_luts is not None else [None] * len(tile_xyz)
        self.tiles = [np.ascontiguousarray(t, dtype=np.float64) if t is not None else np.zeros((1, 3))
                      for t in tile_xyz]
        self.n_tiles = [len(t) if b is None else int(bin_counts[k])
                        for k, (t, b) in enumerate(zip(self.tiles, bin_luts))]
        self.weighted = bool(use_weight_distribution)
        self.width, self.height = int(video_width), int(video_height)
        d = _PlanDesc()
        keep = []
        if dir_table is None:
            axes = _quantiser.axis_trig(self.width, self.height)
            keep.extend(axes)
            d.video_width, d.video_height = self.width, self.height
            d.h_lon_cos, d.h_lon_sin, d.h_lat_sin, d.h_lat_cos = (a.ctypes.data for a in axes)
        else:
            tab = np.ascontiguousarray(dir_table, dtype=np.float64).reshape(-1, 3)
            keep.append(tab)
            d.h_dir_table, d.n_dirs = tab.ctypes.data, len(tab)
        n_arr = np.asarray(self.n_tiles, dtype=np.int32)
        ptrs = (C.c_void_p * len(self.tiles))(*[t.ctypes.data for t in self.tiles])
        hmax = np.asarray([_quantiser.max_entropy(n) if b is None else float(bin_max_entropy[k])
                           for k, (n, b) in enumerate(zip(self.n_tiles, bin_luts))], dtype=np.float64)
        keep.extend([n_arr, ptrs, hmax])
        if any(b is not None for b in bin_luts):
            luts = [None if b is None else np.ascontiguousarray(b, dtype=np.uint16).reshape(-1) for b in bin_luts]
            lut_ptrs = (C.c_void_p * len(luts))(*[None if b is None else b.ctypes.data for b in luts])
            norm = np.asarray([n if b is None else int(bin_norm_tiles[k])
                               for k, (n, b) in enumerate(zip(self.n_tiles, bin_luts))], dtype=np.int32)
            keep.extend([luts, lut_ptrs, norm])
            d.h_bin_lut = C.cast(lut_ptrs, C.c_void_p)
            d.n_norm_tiles = norm.ctypes.data
        d.n_lattices = len(self.tiles)
        d.n_tiles = n_arr.ctypes.data
        d.h_tiles = C.cast(ptrs, C.c_void_p)
        d.h_max_entropy = hmax.ctypes.data
        d.fov_angle = float(fov_angle)
        d.max_angular_distance = float(np.radians(fov_angle / 2.0))
        d.power_factor = float(power_factor)
        d.use_weight_distribution = int(self.weighted)
        h = C.c_void_p()
        _check(self.lib, self.lib.vet_plan_create(engine.handle, C.byref(d), C.byref(h)))
        self.handle = h
        self.n_dirs = int(self.lib.vet_plan_n_dirs(h))

    def close(self):
        if getattr(self, "handle", None):
            self.lib.vet_plan_destroy(self.handle)
            self.handle = None

    def __del__(self):  # plans are small; free device tables with the Python object
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def set_table_policy(self, policy: int):
        """0 auto, 1 always use the direction weight table, -1 never (brute-force sweep)."""
        _check(self.lib, self.lib.vet_plan_set_table_policy(self.handle, int(policy)))

    def set_raw_weights(self, on: bool = True):
        """Diagnostic: ``weights`` = the formulation's own histogram (table / sweep resolution) instead of the reference's
        values from the weights-only pass of the precise sweep (include/vet.h: vet_plan_set_raw_weights)."""
        _check(self.lib, self.lib.vet_plan_set_raw_weights(self.handle, 1 if on else 0))

    def table_stride(self, lattice: int = 0) -> int:
        return int(self.lib.vet_plan_table_stride(self.handle, lattice))

    def table_rows(self) -> int:
        """Rows of the plan's weight tables: distinct directions up to the lattices' mirror symmetry (0 before a table exists)."""
        return int(self.lib.vet_plan_table_rows(self.handle))

    def last_formulation(self, lattice: int = 0) -> str:
        """'table' | 'sweep' | 'precise' | 'ftable' of the last weighted call ('' before any)."""
        return {0: "table", 1: "sweep", 2: "precise", 3: "ftable"}.get(int(self.lib.vet_plan_last_formulation(self.handle, lattice)), "")

    def error_bounds(self, lattice: int = 0):
        """(table bound, sweep bound): proven worst-case relative entropy error of the integer formulations."""
        a, b = C.c_double(), C.c_double()
        _check(self.lib, self.lib.vet_plan_error_bounds(self.handle, lattice, C.byref(a), C.byref(b)))
        return a.value, b.value

    # --- parity hooks ---------------------------------------------------------
    def read_dirs(self) -> np.ndarray:
        out = np.empty((self.n_dirs, 3), dtype=np.float64)
        _check(self.lib, self.lib.vet_plan_read_dirs(self.handle, _ptr(out)))
        return out

    def read_nearest(self, lattice: int = 0) -> np.ndarray:
        out = np.empty(self.n_dirs, dtype=np.int32)
        _check(self.lib, self.lib.vet_plan_read_nearest(self.handle, lattice, _ptr(out)))
        return out

    # --- host-buffer runs (what the analyzers use) ---------------------------
    @staticmethod
    def _samples(mu, mv, ids):
        if ids is not None:
            ids = np.ascontiguousarray(ids, dtype=np.int32)
            return None, None, ids, ids.shape
        mu = np.ascontiguousarray(mu, dtype=np.float64)
        mv = np.ascontiguousarray(mv, dtype=np.float64)
        if mu.shape != mv.shape or mu.ndim != 2:
            raise ValueError("mu and mv must be [n_frames, n_users] arrays of equal shape")
        return mu, mv, None, mu.shape

    def spatial(self, mu=None, mv=None, ids=None, want_assign=True, want_weights=False, check=True):
        """Returns dict(entropy[T], assign[T,U]|None, weights[T,n0]|None, present[T], code)."""
        mu, mv, ids, (T, U) = self._samples(mu, mv, ids)
        ent = np.empty(T, dtype=np.float64)
        assign = np.empty((T, U), dtype=np.int32) if want_assign else None
        weights = np.empty((T, self.n_tiles[0]), dtype=np.float64) if want_weights else None
        present = np.empty(T, dtype=np.int32)
        rc = self.lib.vet_spatial_entropy_host(self.handle, _ptr(mu), _ptr(mv), _ptr(ids), U, T, _ptr(ent),
                                               _ptr(assign), _ptr(weights), _ptr(present))
        if rc not in (VET_OK, VET_ERR_EMPTY, VET_ERR_RANGE) or (check and rc != VET_OK):
            _check(self.lib, rc)
        return dict(entropy=ent, assign=assign, weights=weights, present=present, code=rc)

    def transition(self, mu=None, mv=None, ids=None, want_pairs=True, want_srccount=False, check=True):
        """Returns dict(entropy[T-1], pairs[T-1,U,2]|None, srccount[T-1,n0]|None, common[T-1], code)."""
        mu, mv, ids, (T, U) = self._samples(mu, mv, ids)
        R = max(T - 1, 0)
        ent = np.empty(R, dtype=np.float64)
        pairs = np.empty((R, U, 2), dtype=np.int32) if want_pairs else None
        src = np.empty((R, self.n_tiles[0]), dtype=np.int32) if want_srccount else None
        common = np.empty(R, dtype=np.int32)
        rc = self.lib.vet_transition_entropy_host(self.handle, _ptr(mu), _ptr(mv), _ptr(ids), U, T, _ptr(ent),
                                                  _ptr(pairs), _ptr(src), _ptr(common))
        if rc not in (VET_OK, VET_ERR_EMPTY, VET_ERR_RANGE) or (check and rc != VET_OK):
            _check(self.lib, rc)
        return dict(entropy=ent, pairs=pairs, srccount=src, common=common, code=rc)

    def spatial_resident(self, mu=None, mv=None, ids=None, check=True):
        """Like ``spatial`` but only entropy[T] and present[T] come back; the tile assignments and weights stay
        on the device in ``result`` (a ``DeviceResult``) and are fetched by row on demand."""
        mu, mv, ids, (T, U) = self._samples(mu, mv, ids)
        ent = np.empty(T, dtype=np.float64)
        present = np.empty(T, dtype=np.int32)
        h = C.c_void_p()
        rc = self.lib.vet_spatial_entropy_host_resident(self.handle, _ptr(mu), _ptr(mv), _ptr(ids), U, T, _ptr(ent),
                                                        _ptr(present), C.byref(h))
        result = DeviceResult(self.lib, h, T, [(U,), (self.n_tiles[0],)], [np.int32, np.float64], self.engine) if h.value else None
        if rc not in (VET_OK, VET_ERR_EMPTY, VET_ERR_RANGE) or (check and rc != VET_OK):
            if result is not None:
                result.close()
            _check(self.lib, rc)
        return dict(entropy=ent, present=present, result=result, code=rc)

    def transition_resident(self, mu=None, mv=None, ids=None, check=True):
        """Like ``transition`` with the tile pairs and source-tile counts kept on the device."""
        mu, mv, ids, (T, U) = self._samples(mu, mv, ids)
        R = max(T - 1, 0)
        ent = np.empty(R, dtype=np.float64)
        common = np.empty(R, dtype=np.int32)
        h = C.c_void_p()
        rc = self.lib.vet_transition_entropy_host_resident(self.handle, _ptr(mu), _ptr(mv), _ptr(ids), U, T, _ptr(ent),
                                                           _ptr(common), C.byref(h))
        result = DeviceResult(self.lib, h, R, [(U, 2), (self.n_tiles[0],)], [np.int32, np.int32], self.engine) if h.value else None
        if rc not in (VET_OK, VET_ERR_EMPTY, VET_ERR_RANGE) or (check and rc != VET_OK):
            if result is not None:
                result.close()
            _check(self.lib, rc)
        return dict(entropy=ent, common=common, result=result, code=rc)

    def spatial_batch(self, videos, want_assign=False, check=True):
        """Many videos, one launch.  ``videos``: sequence of (mu[T,U], mv[T,U]).  Returns a list of
        dict(entropy[T], assign[T,U]|None, present[T]) in the same order."""
        mus = [np.ascontiguousarray(m, dtype=np.float64) for m, _ in videos]
        mvs = [np.ascontiguousarray(v, dtype=np.float64) for _, v in videos]
        T = np.asarray([m.shape[0] for m in mus], dtype=np.int32)
        U = np.asarray([m.shape[1] for m in mus], dtype=np.int32)
        mu = np.concatenate([m.ravel() for m in mus])
        mv = np.concatenate([v.ravel() for v in mvs])
        ent = np.empty(int(T.sum()), dtype=np.float64)
        present = np.empty(int(T.sum()), dtype=np.int32)
        assign = np.empty(mu.size, dtype=np.int32) if want_assign else None
        rc = self.lib.vet_spatial_entropy_batch_host(self.handle, len(mus), _ptr(U), _ptr(T), _ptr(mu), _ptr(mv),
                                                     _ptr(ent), _ptr(assign), _ptr(present))
        if rc not in (VET_OK, VET_ERR_EMPTY, VET_ERR_RANGE) or (check and rc != VET_OK):
            _check(self.lib, rc)
        out, so, ro = [], 0, 0
        for t, u in zip(T.tolist(), U.tolist()):
            out.append(dict(entropy=ent[ro:ro + t], present=present[ro:ro + t],
                            assign=assign[so:so + t * u].reshape(t, u) if want_assign else None))
            so += t * u
            ro += t
        return out

    def transition_batch(self, videos, want_pairs=False, check=True):
        """Transition mode, many videos, one launch per lattice.  ``videos``: sequence of (mu[T,U], mv[T,U]), T >= 2.
        Returns a list of dict(entropy[T-1], pairs[T-1,U,2]|None, common[T-1]) in the same order."""
        mus = [np.ascontiguousarray(m, dtype=np.float64) for m, _ in videos]
        mvs = [np.ascontiguousarray(v, dtype=np.float64) for _, v in videos]
        T = np.asarray([m.shape[0] for m in mus], dtype=np.int32)
        U = np.asarray([m.shape[1] for m in mus], dtype=np.int32)
        mu = np.concatenate([m.ravel() for m in mus])
        mv = np.concatenate([v.ravel() for v in mvs])
        R = int((T - 1).sum())
        ent = np.empty(R, dtype=np.float64)
        common = np.empty(R, dtype=np.int32)
        pairs = np.empty(int(((T - 1) * U).sum()) * 2, dtype=np.int32) if want_pairs else None
        rc = self.lib.vet_transition_entropy_batch_host(self.handle, len(mus), _ptr(U), _ptr(T), _ptr(mu), _ptr(mv),
                                                        _ptr(ent), _ptr(pairs), _ptr(common))
        if rc not in (VET_OK, VET_ERR_EMPTY, VET_ERR_RANGE) or (check and rc != VET_OK):
            _check(self.lib, rc)
        out, po, ro = [], 0, 0
        for t, u in zip(T.tolist(), U.tolist()):
            out.append(dict(entropy=ent[ro:ro + t - 1], common=common[ro:ro + t - 1],
                            pairs=pairs[po:po + (t - 1) * u * 2].reshape(t - 1, u, 2) if want_pairs else None))
            po += (t - 1) * u * 2
            ro += t - 1
        return out

    def transition_batch_device(self, videos, d_status: int = 0, stream=None):
        """``videos``: ctypes array of ``Video`` (device pointers: d_entropy [T-1], d_assign = pairs, d_present = common)."""
        _check(self.lib, self.lib.vet_transition_entropy_batch(self.handle, len(videos), videos, d_status or None,
                                                               _stream(stream)))

    def spatial_batch_device(self, videos, d_status: int = 0, stream=None):
        """``videos``: ctypes array of ``Video`` (device pointers); asynchronous on ``stream``
        (``None`` = the engine's own stream, 0 = the legacy null stream, else a hipStream_t handle)."""
        _check(self.lib, self.lib.vet_spatial_entropy_batch(self.handle, len(videos), videos, d_status or None,
                                                            _stream(stream)))

    # --- device-pointer runs (inputs resident in HBM; asynchronous on ``stream``) ----
    def spatial_device(self, d_mu: int, d_mv: int, n_users: int, n_frames: int, d_entropy: int, d_assign: int = 0,
                       d_weights: int = 0, d_present: int = 0, d_status: int = 0, stream=None):
        _check(self.lib, self.lib.vet_spatial_entropy(self.handle, d_mu, d_mv, n_users, n_frames, d_entropy,
                                                      d_assign or None, d_weights or None, d_present or None,
                                                      d_status or None, _stream(stream)))

    def transition_device(self, d_mu: int, d_mv: int, n_users: int, n_frames: int, d_entropy: int, d_pairs: int = 0,
                          d_srccount: int = 0, d_common: int = 0, d_status: int = 0, stream=None):
        _check(self.lib, self.lib.vet_transition_entropy(self.handle, d_mu, d_mv, n_users, n_frames, d_entropy,
                                                         d_pairs or None, d_srccount or None, d_common or None,
                                                         d_status or None, _stream(stream)))
