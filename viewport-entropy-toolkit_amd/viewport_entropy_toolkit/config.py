"""Analyzer configuration (drop-in for the reference's ``config`` module, config.py:23-82).

Same defaults, field names, validation errors and the ``mkdir`` side effect of
``AnalyzerConfig.__post_init__``; ``EntropyConfig`` and ``VisualizationConfig`` are
importable from here as in the reference.
"""

from __future__ import annotations

from dataclasses import dataclass, field
from pathlib import Path
from typing import List

from .utilities.entropy_utils import EntropyConfig
from .utilities.visualization_utils import VisualizationConfig

DEFAULT_VIDEO_DIMENSIONS = {"width": 100, "height": 200}
DEFAULT_TILE_COUNTS = [20, 50, 100, 250, 1000]
DEFAULT_OUTPUT_FORMATS = {"video": ".mp4", "data": ".csv", "plot": ".png"}


@dataclass
class AnalyzerConfig:
    """Video size, lattice sizes, output directory and nested entropy / plot settings."""

    video_width: int = DEFAULT_VIDEO_DIMENSIONS["width"]
    video_height: int = DEFAULT_VIDEO_DIMENSIONS["height"]
    tile_counts: List[int] = field(default_factory=lambda: DEFAULT_TILE_COUNTS)
    output_dir: Path = Path("output")
    entropy_config: EntropyConfig = field(default_factory=EntropyConfig)
    visualization_config: VisualizationConfig = field(default_factory=VisualizationConfig)

    def __post_init__(self) -> None:
        if self.video_width <= 0 or self.video_height <= 0:
            raise ValueError("Video dimensions must be positive")
        if not self.tile_counts:
            raise ValueError("Must specify at least one tile count")
        if any(count <= 0 for count in self.tile_counts):
            raise ValueError("Tile counts must be positive")
        self.output_dir = Path(self.output_dir)
        self.output_dir.mkdir(parents=True, exist_ok=True)

    def get_output_path(self, base_name: str, extension: str) -> Path:
        return self.output_dir / f"{base_name}{extension}"


@dataclass
class NaiveAnalyzerConfig:
    """Configuration of the latitude-longitude grid analyzer (reference config.py:84-126):
    ``tile_height`` must divide 180 and ``tile_width`` 360 (degrees)."""

    video_width: int = DEFAULT_VIDEO_DIMENSIONS["width"]
    video_height: int = DEFAULT_VIDEO_DIMENSIONS["height"]
    output_dir: Path = Path("output")
    entropy_config: EntropyConfig = field(default_factory=EntropyConfig)
    visualization_config: VisualizationConfig = field(default_factory=VisualizationConfig)
    tile_width: int = -1
    tile_height: int = -1

    def __post_init__(self) -> None:
        if self.video_width <= 0 or self.video_height <= 0:
            raise ValueError("Video dimensions must be positive")
        if not self.tile_height or not self.tile_width:
            raise ValueError("Must specify both tile_height and tile_width")
        self.output_dir = Path(self.output_dir)
        self.output_dir.mkdir(parents=True, exist_ok=True)

    def get_output_path(self, base_name: str, extension: str) -> Path:
        return self.output_dir / f"{base_name}{extension}"
