"""Analyzers: per-frame spatial and transition entropy on the MI355X engine."""

from .spatial_entropy import SpatialEntropyAnalyzer
from .transition_entropy import TransitionEntropyAnalyzer
from .naive_spatial_entropy import NaiveSpatialEntropyAnalyzer

__all__ = ["SpatialEntropyAnalyzer", "TransitionEntropyAnalyzer", "NaiveSpatialEntropyAnalyzer"]
