"""Analyzers: per-frame spatial and transition entropy on the MI355X engine."""

from .spatial_entropy import SpatialEntropyAnalyzer
from .transition_entropy import TransitionEntropyAnalyzer

__all__ = ["SpatialEntropyAnalyzer", "TransitionEntropyAnalyzer"]
