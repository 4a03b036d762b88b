"""Analyzer configuration objects.

API-compatible with the reference's ``config`` module (same dataclass names, fields, defaults,
validation errors and the directory-creating side effect of construction; reference
config.py:23-126): ``AnalyzerConfig`` for the Fibonacci-lattice analyzers and
``NaiveAnalyzerConfig`` for the latitude-longitude grid analyzer.  ``EntropyConfig`` and
``VisualizationConfig`` are re-exported here because callers import them from this module.
"""

from __future__ import annotations

import dataclasses
import pathlib
from typing import List

from .utilities.entropy_utils import EntropyConfig
from .utilities.visualization_utils import VisualizationConfig

DEFAULT_VIDEO_DIMENSIONS = dict(width=100, height=200)
DEFAULT_TILE_COUNTS = [20, 50, 100, 250, 1000]
DEFAULT_OUTPUT_FORMATS = dict(video=".mp4", data=".csv", plot=".png")


class _OutputLocation:
    """Shared behaviour of both configurations: positive video size, output directory that
    exists after construction, output file naming."""

    def _prepare(self) -> None:
        if min(self.video_width, self.video_height) <= 0:
            raise ValueError("Video dimensions must be positive")
        self.output_dir = pathlib.Path(self.output_dir)
        self.output_dir.mkdir(parents=True, exist_ok=True)

    def get_output_path(self, base_name: str, extension: str) -> pathlib.Path:
        return self.output_dir / (base_name + extension)


@dataclasses.dataclass
class AnalyzerConfig(_OutputLocation):
    """Video size, the lattice sizes whose entropies are averaged, output directory and the
    nested entropy / plot settings."""

    video_width: int = DEFAULT_VIDEO_DIMENSIONS["width"]
    video_height: int = DEFAULT_VIDEO_DIMENSIONS["height"]
    tile_counts: List[int] = dataclasses.field(default_factory=lambda: DEFAULT_TILE_COUNTS)
    output_dir: pathlib.Path = pathlib.Path("output")
    entropy_config: EntropyConfig = dataclasses.field(default_factory=EntropyConfig)
    visualization_config: VisualizationConfig = dataclasses.field(default_factory=VisualizationConfig)

    def __post_init__(self) -> None:
        if min(self.video_width, self.video_height) <= 0:
            raise ValueError("Video dimensions must be positive")
        if len(self.tile_counts) == 0:
            raise ValueError("Must specify at least one tile count")
        if min(self.tile_counts) <= 0:
            raise ValueError("Tile counts must be positive")
        self._prepare()


@dataclasses.dataclass
class NaiveAnalyzerConfig(_OutputLocation):
    """Latitude-longitude grid analyzer: ``tile_height`` must divide 180 and ``tile_width`` 360
    (degrees); both default to -1 (unset) like the reference."""

    video_width: int = DEFAULT_VIDEO_DIMENSIONS["width"]
    video_height: int = DEFAULT_VIDEO_DIMENSIONS["height"]
    output_dir: pathlib.Path = pathlib.Path("output")
    entropy_config: EntropyConfig = dataclasses.field(default_factory=EntropyConfig)
    visualization_config: VisualizationConfig = dataclasses.field(default_factory=VisualizationConfig)
    tile_width: int = -1
    tile_height: int = -1

    def __post_init__(self) -> None:
        if min(self.video_width, self.video_height) <= 0:
            raise ValueError("Video dimensions must be positive")
        if not (self.tile_height and self.tile_width):
            raise ValueError("Must specify both tile_height and tile_width")
        self._prepare()
