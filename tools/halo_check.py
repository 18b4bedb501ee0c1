"""Source-tile aggregation (agg_halo_kernel) against the per-edge gather (agg_kernel): bit equality and time.

    python tools/halo_check.py [--levels 3,5] [--F 64] [--B 64] [--iters 30]

Development tool (GPU): the mesh graph of the named levels is renumbered tile by tile (mesh.tile_order) and
aggregated forward and transposed by both kernels (GCL_AGG_HALO switches per call)."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graphcast_lite_amd import hip  # noqa: E402
from graphcast_lite_amd.create_graphs import create_processing_graph  # noqa: E402
from graphcast_lite_amd.mesh import get_hierarchy_of_triangular_meshes_for_sphere, tile_order  # noqa: E402


def timeit(fn, iters, reps=10):
    """Median / min time of one call in us: `reps` calls back to back between two events (one call between two
    events also times the host's gap between the first event and the launch: 5-8 us from Python)."""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--levels", default="3,5")
    ap.add_argument("--F", type=int, default=64)
    ap.add_argument("--B", type=int, default=64)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--leaf", type=int, default=64)
    ap.add_argument("--no-renumber", action="store_true")
    ap.add_argument("--degree-sort", action="store_true")
    args = ap.parse_args()
    levels = [int(x) for x in args.levels.split(",")]
    dev = torch.device("cuda:0")
    meshes = get_hierarchy_of_triangular_meshes_for_sphere(splits=max(levels))
    ei = create_processing_graph(meshes, levels)
    M = len(meshes[-1].vertices)
    if not args.no_renumber:
        deg = np.bincount(ei[1].numpy(), minlength=M) if args.degree_sort else None
        order = tile_order(meshes[-1].vertices, args.leaf, degree=deg)
        pos = np.empty(M, dtype=np.int64)
        pos[order] = np.arange(M)
        ei = torch.from_numpy(pos)[ei]
    g = hip.Graph(ei, M, hip.GRAPH_GCN)
    print(f"# mesh {levels}: M={M} E'={g.e} halo fwd {g.halo_info(False, 64)} / {g.halo_info(False, 32)} "
          f"transposed {g.halo_info(True, 64)} / {g.halo_info(True, 32)}")
    B, F = args.B, args.F
    gen = torch.Generator().manual_seed(0)
    h = torch.randn(B, M, F, generator=gen).to(dev)
    bias = torch.randn(F, generator=gen).to(dev)
    per = 4 * M * 2 * F + 4 * g.e + 4 * (M + 1) + 4 * M
    tmp = torch.empty_like(h)
    us, mn = timeit(lambda: tmp.copy_(h), args.iters)
    print(f"torch copy of h ({h.numel() * 4 / 1e6:.0f} MB read + write): {us:8.1f} us (min {mn:8.1f}) = {2 * h.numel() * 4 / us / 1e3:8.1f} GB/s")
    for tr in (False, True):
        outs = {}
        for mode in ("0", "1"):
            os.environ["GCL_AGG_HALO"] = mode
            out = torch.full((B, M, F), float("nan"), device=dev)
            hip.aggregate(g, h, None if tr else bias, transpose=tr, out=out)
            torch.cuda.synchronize()
            outs[mode] = out
            us, mn = timeit(lambda: hip.aggregate(g, h, None if tr else bias, transpose=tr, out=out), args.iters)
            print(f"{'transposed' if tr else 'forward   '} halo={mode}: {us:8.1f} us (min {mn:8.1f})  "
                  f"{B * per / us / 1e3:8.1f} GB/s algorithmic = {B * per / us / 8e6 * 100:5.1f} % of 8 TB/s", flush=True)
        same = torch.equal(outs["0"], outs["1"])
        bits = (outs["0"].view(torch.int32) != outs["1"].view(torch.int32)).sum().item()
        print(f"  equal: {same}; elements whose bits differ: {bits}; max |diff| {(outs['0'] - outs['1']).abs().max().item():.3e}")
        assert same and bits == 0


if __name__ == "__main__":
    main()
