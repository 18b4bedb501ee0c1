"""Per-kernel micro-benchmark at the BASELINE.json shapes (HIP events, median of N launches).

    python tools/kbench.py [--config baseline|wb2] [--iters 20]

Prints for every C-ABI kernel on the training path: time, algorithmic GB/s and TFLOP/s, so a
kernel can be read against its roofline (HBM 8 TB/s spec / 6.3 measured; fp32 MFMA 157 TFLOP/s).
Development tool: not part of the product path, the tests or bench.py.
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphcast_lite_amd import hip  # noqa: E402
from graphcast_lite_amd.experiments import GRID, experiment  # noqa: E402
from graphcast_lite_amd.models import WeatherPrediction  # noqa: E402


def timeit(fn, iters, reps=5):
    """Median / min time of one call in us: `reps` calls back to back between two events (a single call between
    two events also times the host's gap between the first event and the launch, 5-8 us from Python)."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3 / reps)
    return float(np.median(ts)), float(np.min(ts))


def row(name, us, mn, bytes_, flops):
    print(f"{name:44s} {us:9.1f} us (min {mn:8.1f})  {bytes_ / us / 1e3:8.1f} GB/s  {flops / us / 1e6:7.2f} TFLOP/s", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="baseline")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--batch", type=int, default=0)
    ap.add_argument("--ring", type=int, default=1, help="rotate the aggregation / layer launches over this many distinct input and "
                    "output buffers (>= 3 at cfg A keeps the 256 MB Infinity Cache from serving a repeated launch its own previous input)")
    ap.add_argument("--only", default="", help="comma list of: copy,linear,agg,gcn,gat,norm,misc (default all but gat)")
    args = ap.parse_args()
    name = "wb2_512x256_19f_ar" if args.config.startswith("wb2") else args.config
    dev = torch.device("cuda:0")
    cfg = experiment(name)
    nlat, nlon = GRID[name]
    m = WeatherPrediction((np.linspace(-90, 90, nlat), np.linspace(0, 360, nlon, endpoint=False)), cfg.graph,
                          cfg.pipeline, cfg.data, dev)
    B = args.batch or (8 if name.startswith("wb2") else 64)
    G, M = m._num_grid_nodes, m._num_mesh_nodes
    n = G + M
    F = cfg.pipeline.processor.gcn.output_dim
    print(f"# {name}: B={B} G={G} M={M} n={n} F={F}")
    g = torch.Generator().manual_seed(0)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
    only = set(args.only.split(",")) if args.only else {"copy", "linear", "agg", "gcn", "norm", "misc"}

    # stream copy ceiling
    if "copy" in only:
      src, dst = rnd(64 * 1024 * 1024), torch.empty(64 * 1024 * 1024, device=dev)
      us, mn = timeit(lambda: dst.copy_(src), args.iters)
      row("torch copy 256 MiB (read+write)", us, mn, 2 * src.numel() * 4, 0)

    slope = torch.tensor([0.25], device=dev)
    for rows, K, N, tag in () if "linear" not in only else ((B * n, F, F, "enc/dec"), (B * M, F, F, "mesh"), (B * n, cfg.data.num_features_used * 2 + 6, cfg.pipeline.encoder.mlp.mlp_hidden_dims[0], "mlp0")):
        x, W, b = rnd(rows, K), rnd(N, K) * 0.1, rnd(N)
        y = torch.empty(rows, N, device=dev)
        us, mn = timeit(lambda: hip.linear_fwd(x, W, b, slope, out=y), args.iters)
        row(f"linear_fwd {tag} [{rows}x{K}]->{N}", us, mn, 4 * rows * (K + N), 2 * rows * K * N)
        dy = rnd(rows, N)
        ds = torch.zeros(1, device=dev)
        us, mn = timeit(lambda: hip.linear_bwd_dx(dy, W, x, slope, ds), args.iters)
        row(f"linear_bwd_dx {tag} [{rows}x{N}]->{K}", us, mn, 4 * rows * (2 * K + N), 2 * rows * K * N)
        dW, db = torch.empty(N, K, device=dev), torch.empty(N, device=dev)
        us, mn = timeit(lambda: hip.linear_bwd_dw(dy, x, slope, dW, db, False), args.iters)
        row(f"linear_bwd_dw {tag} [{rows}x{N}]^T[{rows}x{K}]", us, mn, 4 * rows * (K + N), 2 * rows * K * N)
        cs = torch.empty(K, device=dev)
        us, mn = timeit(lambda: hip.linear_bwd_all(dy, W, x, slope, ds, dW, db, cs, False), args.iters)
        row(f"linear_bwd_all {tag} (dx+dW+db+colsum)", us, mn, 4 * rows * (2 * K + N), 4 * rows * K * N)

    from graphcast_lite_amd.models import _graphs
    for ei, nn_, tag in () if "agg" not in only else ((m.processor_graph(), M, "mesh E_M (tile order)"), (m.encoding_graph, n, "enc E_G2M"), (m.decoding_graph, n, "dec E_M2G")):
        gr = _graphs.get(ei, nn_, hip.GRAPH_GCN)
        bias = rnd(F)
        hs, outs, k = [rnd(B, nn_, F) for _ in range(args.ring)], [torch.empty(B, nn_, F, device=dev) for _ in range(args.ring)], [0]

        def agg(transpose):
            k[0] = (k[0] + 1) % args.ring
            hip.aggregate(gr, hs[k[0]], None if transpose else bias, transpose=transpose, out=outs[k[0]])
        per = 4 * nn_ * 2 * F + 4 * gr.e + 4 * (nn_ + 1) + 4 * nn_
        us, mn = timeit(lambda: agg(False), args.iters)
        row(f"aggregate fwd {tag} n={nn_} E'={gr.e} F={F}", us, mn, B * per, 2 * B * gr.e * F)
        us, mn = timeit(lambda: agg(True), args.iters)
        row(f"aggregate bwd {tag} (transpose)", us, mn, B * per, 2 * B * gr.e * F)

    # one-kernel GCNConv layer (aggregate-first): the same per-layer algorithmic bytes as the aggregation
    for ei, nn_, tag in () if "gcn" not in only else ((m.processor_graph(), M, "mesh E_M (tile order)"), (m.encoding_graph, n, "enc E_G2M"), (m.decoding_graph, n, "dec E_M2G")):
        gr = _graphs.get(ei, nn_, hip.GRAPH_GCN)
        W, bias = rnd(F, F) * 0.1, rnd(F)
        xs, outs, k = [rnd(B, nn_, F) for _ in range(args.ring)], [torch.empty(B, nn_, F, device=dev) for _ in range(args.ring)], [0]

        def layer(act):
            k[0] = (k[0] + 1) % args.ring
            hip.gcn_layer_fwd(gr, xs[k[0]], act, slope if act else None, W, bias, out=outs[k[0]])
        per = 4 * nn_ * 2 * F + 4 * gr.e + 4 * (nn_ + 1) + 4 * nn_
        for act, an in ((hip.ACT_NONE, "none"), (hip.ACT_PRELU, "prelu")):
            us, mn = timeit(lambda: layer(act), args.iters)
            row(f"gcn_layer_fwd {tag} act={an} F={F}", us, mn, B * per, 2 * B * nn_ * F * F + 2 * B * gr.e * F)

    # GATConv / SparseGATConv attention aggregation on the mesh graph (H = 1 head of C = F channels, as configs[2]/[4])
    if "gat" in only:
        H, C = 1, F
        from graphcast_lite_amd.mesh import tile_order
        order = torch.from_numpy(np.ascontiguousarray(tile_order(m._finest_mesh.vertices, 64, degree=np.bincount(m.processing_graph[1].cpu().numpy(), minlength=M))))
        posm = torch.empty(M, dtype=torch.int64)
        posm[order] = torch.arange(M)
        gr = hip.Graph(posm[m.processing_graph.cpu()], M, hip.GRAPH_GAT)  # mesh rows in tile order, as the model runs a GAT processor
        h = rnd(B, M, H * C)
        a_s, a_d, bias = rnd(H * C) * 0.3, rnd(H * C) * 0.3, rnd(C)
        # per sample: read h at every edge end is served from cache; algorithmic = h once + y once + alpha + indices
        per = 4 * M * (H * C + C) + 4 * gr.e * H + 4 * gr.e + 4 * (M + 1)
        us, mn = timeit(lambda: hip.gat_fwd(gr, h, a_s, a_d, bias, H, C), args.iters)
        row(f"gat_fwd mesh n={M} E'={gr.e} H={H} C={C}", us, mn, B * per, 2 * B * gr.e * H * C * 2)
        y, s_src, s_dst, alpha = hip.gat_fwd(gr, h, a_s, a_d, bias, H, C)
        dy = rnd(B, M, C)
        d_as, d_ad, d_b = torch.empty(H * C, device=dev), torch.empty(H * C, device=dev), torch.empty(C, device=dev)
        us, mn = timeit(lambda: hip.gat_bwd(gr, dy, h, a_s, a_d, s_src, s_dst, alpha, d_as, d_ad, d_b, False, H, C), args.iters)
        row(f"gat_bwd mesh (dh, d_att, d_bias)", us, mn, B * (per + 4 * M * H * C + 4 * M * C), 2 * B * gr.e * H * C * 4)

    if "norm" not in only:
        return
    rows = B * n
    x, gm, bt = rnd(rows, F), rnd(F), rnd(F)
    us, mn = timeit(lambda: hip.layernorm_fwd(x, gm, bt), args.iters)
    row(f"layernorm_fwd [{rows}x{F}]", us, mn, 4 * rows * 2 * F, 0)
    y, st = hip.layernorm_fwd(x, gm, bt)
    dg, dbt = torch.empty(F, device=dev), torch.empty(F, device=dev)
    us, mn = timeit(lambda: hip.layernorm_bwd(x, x, gm, st, dg, dbt, False), args.iters)
    row(f"layernorm_bwd [{rows}x{F}]", us, mn, 4 * rows * 3 * F, 0)
    o = torch.empty(F, device=dev)
    us, mn = timeit(lambda: hip.colsum(x, o, False), args.iters)
    row(f"colsum [{rows}x{F}]", us, mn, 4 * rows * F, 0)
    X = rnd(B, G, cfg.data.num_features_used * 2)
    us, mn = timeit(lambda: hip.assemble_input(X, m.init_grid_features, m.init_mesh_features), args.iters)
    row("assemble_input", us, mn, 4 * B * n * (X.shape[-1] + 6), 0)


if __name__ == "__main__":
    main()
