"""Shared analyzer shell: ingest, plan cache, visualisation outputs, run_analysis.

Mirrors the reference's analyzer classes (analyzers/spatial_entropy.py:40-253,
analyzers/transition_entropy.py) — same constructor, methods, cached attributes, output file
names and error conventions — with ``compute_entropy`` replaced by one call into the HIP
engine per video.
"""

from __future__ import annotations

import logging
import os
import time
from datetime import datetime
from pathlib import Path
from typing import Dict, List, Optional

import numpy as np
import pandas as pd

from .. import _ingest, _native, _quantiser
from ..config import AnalyzerConfig, DEFAULT_OUTPUT_FORMATS
from ..data_types import Vector, ValidationError
from ..utilities.data_utils import format_trajectory_data, generate_fibonacci_lattice
from ..utilities.visualization_utils import save_graph


class _DataCache(dict):
    """``{'trajectory_data', 'points', 'vectors'}`` of the reference, every entry built on first
    access from the cleaned samples (the engine itself only needs the dense arrays)."""

    def __init__(self, samples, width, height):
        super().__init__()
        self._samples, self._dims = samples, (width, height)
        self._lazy = {"trajectory_data", "points", "vectors"}
        self.user_vectors = False

    def _fill(self, key):
        if "trajectory_data" in self._lazy:
            self._lazy.discard("trajectory_data")
            dict.__setitem__(self, "trajectory_data",
                             [(name, _ingest.frame_of_samples(labels, t, a, b, *self._dims))
                              for labels, t, a, b, name in self._samples])
        if key != "trajectory_data" and key in self._lazy:
            self._lazy -= {"points", "vectors"}
            copies = [(name, df.copy()) for name, df in dict.__getitem__(self, "trajectory_data")]
            points, vectors = format_trajectory_data(copies)
            dict.__setitem__(self, "points", points)
            dict.__setitem__(self, "vectors", vectors)

    def __getitem__(self, key):
        if key in self._lazy:
            self._fill(key)
        return dict.__getitem__(self, key)

    def __setitem__(self, key, value):
        self._lazy.discard(key)
        if key == "vectors":
            self.user_vectors = True      # caller replaced the frame table: honour it
        dict.__setitem__(self, key, value)

    def get(self, key, default=None):
        try:
            return self[key]
        except KeyError:
            return default

    def __contains__(self, key):
        return key in self._lazy or dict.__contains__(self, key)

    def __bool__(self):
        return True

    def keys(self):
        return list(dict.keys(self)) + sorted(self._lazy)


class _EntropyAnalyzerBase:
    _logger = logging.getLogger(__name__)

    def __init__(self, config: Optional[AnalyzerConfig] = None):
        self.config = config or AnalyzerConfig()
        self.plot_manager = None          # matplotlib scatter animation is outside this engine
        self._data_cache: dict = {}
        self._entropy_results: Optional[pd.DataFrame] = None
        self._fibonacci_vectors: Dict[int, List[Vector]] = {
            count: generate_fibonacci_lattice(count) for count in self.config.tile_counts
        }
        self._dense = None                # (frame_times[T], mu[T,U], mv[T,U], user names)
        self._plan: Optional[_native.Plan] = None
        self._plan_key = None
        self.last_timing: Dict[str, float] = {}   # seconds / rates of the last ingest and compute

    # ------------------------------------------------------------------ ingest
    def process_directory(self, directory: Path) -> None:
        """Reads every ``*.csv`` of ``directory`` (one user each, in glob order)."""
        directory = Path(directory)
        if not directory.exists():
            raise FileNotFoundError(f"Directory not found: {directory}")
        t_start = time.perf_counter()
        try:
            files = list(directory.glob("*.csv"))          # glob order = user (column) order, as in the reference
            samples = _ingest.read_directory(files, self.config.video_width, self.config.video_height)
            times, mu, mv = _ingest.build_dense([(t, a, b) for _, t, a, b, _ in samples])
            self._dense = (times, mu, mv, [name for *_, name in samples])
            self._data_cache = _DataCache(samples, self.config.video_width, self.config.video_height)
            self._entropy_results = None
            self.last_timing = {"ingest_s": time.perf_counter() - t_start}
        except Exception as e:  # noqa: BLE001
            self._logger.error(f"Error processing directory {directory}: {str(e)}")
            raise ValidationError(f"Failed to process directory: {str(e)}")

    def load_arrays(self, frame_times: np.ndarray, mu: np.ndarray, mv: np.ndarray,
                    user_names: Optional[List[str]] = None) -> None:
        """Engine-native ingest: dense frame-major arrays (NaN = absent) instead of CSV files.

        This IS the ingest step, so the ingest's checks are made here and in the reference's order
        (process_viewport_data, data_utils.py:322-331: 2dmu range, 2dmv range, video dimensions): a sample outside
        [0, 1] raises ``ValidationError`` now, as ``process_directory`` would, and never competes with a compute-time
        error (an empty frame) of ``compute_entropy`` — whichever row either sits in."""
        mu = np.ascontiguousarray(mu, dtype=np.float64)
        mv = np.ascontiguousarray(mv, dtype=np.float64)
        if mu.ndim != 2 or mu.shape != mv.shape or len(frame_times) != mu.shape[0]:
            raise ValidationError("frame_times[T], mu[T,U], mv[T,U] expected")
        _ingest.check_normalized(mu, self.config.video_width)       # NaN = absent passes
        _ingest.check_normalized(mv, self.config.video_height)
        _ingest.check_video_dimensions(self.config.video_width, self.config.video_height)
        names = list(user_names) if user_names is not None else [f"user{u:03d}" for u in range(mu.shape[1])]
        self._dense = (np.asarray(frame_times, dtype=np.float64), mu, mv, names)
        self._data_cache = {"dense": True}
        self._entropy_results = None

    # ------------------------------------------------------------------ engine
    def _get_plan(self, dir_table: Optional[np.ndarray] = None) -> "_native.Plan":
        ec = self.config.entropy_config
        key = (tuple(self.config.tile_counts), self.config.video_width, self.config.video_height,
               ec.fov_angle, ec.power_factor, ec.use_weight_distribution, dir_table is None)
        if dir_table is not None or self._plan is None or self._plan_key != key:
            tiles = [np.array([[v.x, v.y, v.z] for v in self._fibonacci_vectors[c]], dtype=np.float64)
                     for c in self.config.tile_counts]
            plan = _native.Plan(_native.Engine.default(), tiles, ec.fov_angle, ec.power_factor,
                                ec.use_weight_distribution, self.config.video_width, self.config.video_height,
                                dir_table=dir_table)
            if dir_table is not None:
                return plan
            self._plan, self._plan_key = plan, key
        return self._plan

    def _samples(self):
        """Dense samples for the engine: ('grid', times, mu, mv, names) from this analyzer's own
        ingest, or ('ids', times, ids, table, names) for a hand-assigned ``_data_cache['vectors']``."""
        if not self._data_cache:
            raise ValidationError("No data available. Call process_directory first.")
        if self._dense is not None and not getattr(self._data_cache, "user_vectors", False):
            times, mu, mv, names = self._dense
            return "grid", times, mu, mv, names
        vectors_df = self._data_cache["vectors"]
        names = [c for c in vectors_df.columns if c != "time"]
        table: Dict[Vector, int] = {}
        ids = np.full((len(vectors_df), len(names)), -1, dtype=np.int32)
        for j, name in enumerate(names):
            for i, v in enumerate(vectors_df[name]):
                if v is not None:
                    ids[i, j] = table.setdefault(v, len(table))
        xyz = np.array([[v.x, v.y, v.z] for v in table], dtype=np.float64).reshape(-1, 3)
        return "ids", vectors_df["time"].to_numpy(dtype=np.float64), ids, xyz, names

    # ------------------------------------------------------------------ outputs
    def create_visualization(self, base_name: str) -> None:
        """Writes ``{base_name}_graph.png`` and ``{base_name}.csv`` (columns time, entropy).

        The per-frame scatter animation / mp4 of the reference is host matplotlib + ffmpeg
        work outside this engine and is not produced."""
        if self._entropy_results is None:
            raise ValidationError("No entropy results. Call compute_entropy first.")
        try:
            save_graph(entropy_values=self._entropy_results["entropy"], time_values=self._entropy_results["time"],
                       output_path=self.config.get_output_path(f"{base_name}_graph", DEFAULT_OUTPUT_FORMATS["plot"]),
                       config=self.config.visualization_config)
            self._entropy_results[["time", "entropy"]].to_csv(
                self.config.get_output_path(base_name, DEFAULT_OUTPUT_FORMATS["data"]), index=False)
        except Exception as e:  # noqa: BLE001
            self._logger.error(f"Error creating visualization: {str(e)}")
            raise RuntimeError(f"Failed to create visualization: {str(e)}")

    def run_analysis(self, directory: Path, output_prefix: str = "") -> None:
        """process_directory -> compute_entropy -> create_visualization; logs and re-raises."""
        try:
            directory = Path(directory)
            self.process_directory(directory)
            self.compute_entropy()
            timestamp = datetime.now().strftime("%Y%m%d_%H%M%S")
            base_name = f"{directory.stem}_{output_prefix}_{timestamp}"
            self.create_visualization(base_name)
            self._logger.info(f"Analysis completed successfully: {base_name}")
        except Exception as e:  # noqa: BLE001
            self._logger.error(f"Analysis failed: {str(e)}")
            raise

    def _record_compute(self, seconds: float, n_samples: int, n_rows: int) -> None:
        """Throughput counters of the last engine call (samples/s, frames/s)."""
        self.last_timing.update(compute_s=seconds, samples=float(n_samples), frames=float(n_rows),
                                samples_per_s=n_samples / max(seconds, 1e-12),
                                frames_per_s=n_rows / max(seconds, 1e-12))
        self._logger.info("entropy of %d frames x %d samples in %.3f ms (%.3g samples/s)", n_rows, n_samples,
                          seconds * 1e3, n_samples / max(seconds, 1e-12))

    def compute_entropy(self) -> pd.DataFrame:  # pragma: no cover - overridden
        raise NotImplementedError
