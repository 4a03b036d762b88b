"""Per-frame entropy analyzers running on the MI355X engine.

Three drop-in classes share one shell (``_base._EntropyAnalyzerBase``: ingest, plan cache,
outputs) and differ in the kernel family they call: FoV-weighted / nearest-tile spatial entropy,
(t-1 -> t) transition entropy, and the latitude-longitude grid variant.
"""

from . import naive_spatial_entropy, spatial_entropy, transition_entropy

SpatialEntropyAnalyzer = spatial_entropy.SpatialEntropyAnalyzer
TransitionEntropyAnalyzer = transition_entropy.TransitionEntropyAnalyzer
NaiveSpatialEntropyAnalyzer = naive_spatial_entropy.NaiveSpatialEntropyAnalyzer

__all__ = ("SpatialEntropyAnalyzer", "TransitionEntropyAnalyzer", "NaiveSpatialEntropyAnalyzer")
