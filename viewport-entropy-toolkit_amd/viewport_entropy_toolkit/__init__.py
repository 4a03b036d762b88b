"""MI355X-native viewport-entropy engine behind the viewport-entropy-toolkit API.

``import viewport_entropy_toolkit`` exposes the public names of the reference package for the
Fibonacci-lattice spatial / transition entropy path (and the naive lat/lon analyzer), so user
code written against the reference keeps working.  Underneath, ``compute_entropy`` is one call
into hand-written HIP kernels through a ctypes C-ABI (module ``_native``, library
``libvet_hip.so``); there is no CPU implementation to fall back to.

Engine-specific additions live in underscore modules: ``_native`` (C-ABI binding), ``_ingest``
(dense vectorised ingest), ``_dist`` (one video per GPU + single gather), ``_synthetic``
(seeded workloads), ``_results`` (lazy per-frame result views).
"""

from . import data_types as _types
from . import config as _config
from . import analyzers as _analyzers

_EXPORTS = {
    _types: ("Point", "RadialPoint", "Vector", "ValidationError", "SpatialError", "convert_vectors_to_coordinates"),
    _config: ("AnalyzerConfig", "DEFAULT_VIDEO_DIMENSIONS", "DEFAULT_TILE_COUNTS"),
    _analyzers: ("SpatialEntropyAnalyzer", "TransitionEntropyAnalyzer", "NaiveSpatialEntropyAnalyzer"),
}
for _module, _names in _EXPORTS.items():
    for _name in _names:
        globals()[_name] = getattr(_module, _name)

__all__ = [name for names in _EXPORTS.values() for name in names]
__version__ = "1.0.0"          # API level of the reference this package mirrors
del _module, _names, _name
