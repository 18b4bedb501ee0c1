"""Diagnostic: where does a wave of the one-kernel GCN layer spend its cycles?  Needs the stamped build:
    make -C graphcast-lite_amd/csrc STAMPS=1 && GCL_LIB=graphcast-lite_amd/libgcl_hip_stamps.so python tools/stamps_gcn.py
Prints per-phase shares (loop top / metadata / gather / MFMA / store) summed over the waves of the first blocks."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphcast_lite_amd import hip  # noqa: E402
from graphcast_lite_amd.experiments import GRID, experiment  # noqa: E402
from graphcast_lite_amd.models import WeatherPrediction, _graphs  # noqa: E402

dev = torch.device("cuda:0")
cfg = experiment("baseline")
nlat, nlon = GRID["baseline"]
m = WeatherPrediction((np.linspace(-90, 90, nlat), np.linspace(0, 360, nlon, endpoint=False)), cfg.graph, cfg.pipeline, cfg.data, dev)
B, F = 64, 64
L = hip.lib()
L.gcl_debug_read_stamps.argtypes = [C.c_void_p, C.c_int]
for ei, n, tag in ((m.processor_graph(), m._num_mesh_nodes, "mesh"), (m.encoding_graph, m._num_grid_nodes + m._num_mesh_nodes, "enc")):
    gr = _graphs.get(ei, n, hip.GRAPH_GCN)
    x, W, b = torch.randn(B, n, F, device=dev), torch.randn(F, F, device=dev) * 0.1, torch.randn(F, device=dev)
    sl = torch.tensor([0.25], device=dev)
    out = torch.empty(B, n, F, device=dev)
    for _ in range(3):
        hip.gcn_layer_fwd(gr, x, hip.ACT_PRELU, sl, W, b, out=out)
    torch.cuda.synchronize()
    buf = np.zeros(8 * 4096, dtype=np.uint64)
    rc = L.gcl_debug_read_stamps(buf.ctypes.data, buf.size)
    assert rc == 0
    raw = buf.reshape(-1, 8).astype(np.float64)
    if tag == "mesh" and gr.halo_info(False, 64) is not None:  # gcn_halo_fwd_kernel: 7 phases + item count in slot 7
        st = raw[raw[:, 7] > 0]
        names = ["wait+barrier", "sums->At", "barrier", "DMA issue", "split+MFMA", "barrier", "transpose+store"]
        per = st[:, :7] / st[:, 7:8]
        print(f"{tag} (source tiles): items per wave {np.median(st[:, 7]):.0f}; median cycles per item: " +
              ", ".join(f"{nm} {np.median(per[:, i]):.0f}" for i, nm in enumerate(names)) + f"; total {np.median(per.sum(axis=1)):.0f}")
        continue
    st = raw[:256 * 12]
    tot = st[:, :5].sum(axis=1)
    names = ["loop top", "metadata", "gather", "mfma", "store"]
    print(f"{tag}: cycles per wave (median) {np.median(tot):.0f}; shares: " + ", ".join(f"{nm} {st[:, i].sum() / tot.sum():.2f}" for i, nm in enumerate(names)))
    print("   per-phase median cycles per wave:", [int(np.median(st[:, i])) for i in range(5)])
