"""Diagnostic: where does a wave of the pipelined source-tile aggregation spend its cycles?  Needs the stamped build:
    make -C graphcast-lite_amd/csrc STAMPS=1 && GCL_LIB=graphcast-lite_amd/libgcl_hip_stamps.so python tools/stamps_agg.py
Prints median cycles per (tile, sample) item and phase of agg_halo_loop_kernel ."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphcast_lite_amd import hip  # noqa: E402
from graphcast_lite_amd.create_graphs import create_processing_graph  # noqa: E402
from graphcast_lite_amd.mesh import get_hierarchy_of_triangular_meshes_for_sphere, tile_order  # noqa: E402

dev = torch.device("cuda:0")
levels = [3, 5]
meshes = get_hierarchy_of_triangular_meshes_for_sphere(splits=max(levels))
ei = create_processing_graph(meshes, levels)
M = len(meshes[-1].vertices)
order = tile_order(meshes[-1].vertices, 64)
pos = np.empty(M, dtype=np.int64)
pos[order] = np.arange(M)
g = hip.Graph(torch.from_numpy(pos)[ei], M, hip.GRAPH_GCN)
B, F = 64, 64
h = torch.randn(B, M, F, device=dev)
out = torch.empty_like(h)
L = hip.lib()
L.gcl_debug_read_agg_stamps.argtypes = [C.c_void_p, C.c_int]
for _ in range(5):
    hip.aggregate(g, h, None, out=out)
torch.cuda.synchronize()
buf = np.zeros(8 * 4096, dtype=np.uint64)
assert L.gcl_debug_read_agg_stamps(buf.ctypes.data, buf.size) == 0
st = buf.reshape(-1, 8).astype(np.float64)
st = st[st[:, 5] > 0]
names = ["new tile", "dma issue", "landed+barrier", "sums+stores", "free barrier"]
tiles = st[:, 5]
print(f"waves {len(st)}, tiles per wave median {np.median(tiles):.0f}")
for i, nm in enumerate(names):
    per = st[:, i] / tiles
    print(f"  {nm:14s} median {np.median(per):8.0f} cycles per item  (p10 {np.percentile(per, 10):.0f}, p90 {np.percentile(per, 90):.0f})")
tot = st[:, :5].sum(axis=1)
print(f"  total per wave median {np.median(tot):.0f} cycles; per tile {np.median(tot / tiles):.0f}")
