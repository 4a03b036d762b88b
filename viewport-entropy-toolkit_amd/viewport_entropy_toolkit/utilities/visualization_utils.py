"""Visualisation configuration and the entropy-over-time graph.

Rendering is outside the MI355X hot path: only the configuration
dataclass that ``AnalyzerConfig`` embeds (reference visualization_utils.py:32-60) and the
plain matplotlib line graph are provided; the per-frame scatter animation, the ffmpeg video
writer and the pyvista sphere renderers are not part of this engine.
"""

from __future__ import annotations

from dataclasses import dataclass
from pathlib import Path
from typing import Optional, Sequence, Tuple

from ..data_types import ValidationError


@dataclass
class VisualizationConfig:
    """Plot / video parameters (all must be positive)."""

    figure_size: Tuple[int, int] = (12, 6)
    fov_point_size: int = 10
    tile_point_size: int = 40
    fps: int = 10
    dpi: int = 100

    def __post_init__(self) -> None:
        if any(x <= 0 for x in self.figure_size):
            raise ValidationError("Figure dimensions must be positive")
        for value, label in ((self.fov_point_size, "FOV point size"), (self.tile_point_size, "Tile point size"),
                             (self.fps, "FPS"), (self.dpi, "DPI")):
            if value <= 0:
                raise ValidationError(f"{label} must be positive")


def save_graph(entropy_values: Sequence[float], time_values: Sequence[float], output_path: Path,
               config: Optional[VisualizationConfig] = None) -> None:
    """Entropy-over-time line graph as a png (host matplotlib, Agg backend)."""
    if len(entropy_values) != len(time_values):
        raise ValueError("Entropy values length must match time values length!")
    import matplotlib
    matplotlib.use("Agg", force=False)
    import matplotlib.pyplot as plt

    fig = plt.figure(figsize=(10, 6))
    plt.plot(list(time_values), list(entropy_values), marker="o", linestyle="-")
    plt.xlabel("Time")
    plt.ylabel("Entropy")
    plt.title("Entropy over Time")
    plt.grid(True)
    plt.savefig(output_path)
    plt.close(fig)
