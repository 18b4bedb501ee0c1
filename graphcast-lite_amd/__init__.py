"""MI355X-native hot path of graphcast-lite: encode-process-decode GNN forward+backward.

Mirrors the reference's module API (`src/models.py`: MLP, SparseGATConv, GraphLayer, Model,
WeatherPrediction; `src/create_graphs.py`; `src/config.py`) over hand-written gfx950 HIP kernels
reached through the C ABI declared in `include/gcl.h`.  There is no CPU fallback: compute
entry points raise if `libgcl_hip.so` is missing.
"""
__version__ = "0.1.0"
