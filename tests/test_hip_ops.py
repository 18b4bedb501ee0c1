"""Per-kernel parity on the GPU: every C-ABI compute entry point against the CPU oracle
(fp32, tolerance 1e-5 relative unless stated) on seeded inputs, including ragged / odd shapes."""
import numpy as np
import pytest
import torch

from conftest import build_graphs, experiment
from oracle import pyg_ops as P
from oracle import train_step as T

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-5


def rel(a, b):
    """max(Frobenius relative error, element-wise max|diff| / max|ref|): every `rel(..) < tol`
    below bounds BOTH the norm-wise error and the worst single element."""
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    fro = ((a - b).norm() / (b.norm() + 1e-30)).item()
    mx = ((a - b).abs().max() / (b.abs().max() + 1e-30)).item() if b.numel() else 0.0
    return max(fro, mx)


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


@pytest.fixture(scope="module")
def hip(lib_built):
    from graphcast_lite_amd import hip as H

    return H


@pytest.mark.parametrize("rows,Fin,Fout", [(1000, 72, 48), (4096, 64, 64), (257, 48, 33), (130, 44, 128),
                                           (5, 19, 19), (3000, 128, 128), (128, 66, 64), (1, 64, 64)])
@pytest.mark.parametrize("with_slope", [False, True])
def test_linear_fwd(hip, rows, Fin, Fout, with_slope):
    x, W, b = rnd(rows, Fin, seed=1), rnd(Fout, Fin, seed=2, scale=0.2), rnd(Fout, seed=3)
    a = torch.tensor([0.25]) if with_slope else None
    ref = (P.prelu(x, a) if with_slope else x) @ W.t() + b
    y = hip.linear_fwd(x.to(DEV), W.to(DEV), b.to(DEV), a.to(DEV) if with_slope else None)
    assert rel(y, ref) < TOL
    y2 = hip.linear_fwd(x.to(DEV), W.to(DEV), None, None, ld_out=(Fout + 3) // 4 * 4)
    assert rel(y2, x @ W.t()) < TOL


@pytest.mark.parametrize("rows,Fin,Fout", [(4096, 64, 64), (1000, 48, 64), (333, 20, 32), (129, 64, 16), (31, 36, 48)])
def test_x3_linear_fwd_exact_on_integers(hip, rows, Fin, Fout):
    """The K, N <= 64 forward runs on the bf16 matrix pipe with 3-way operand splitting (csrc/x3.h).  Small
    integers make every product and partial sum exact in fp32, so ANY error in the operand maps, the piece
    images or the transposed store shows up as a wrong integer: the result must be bit-equal."""
    g = torch.Generator().manual_seed(5)
    x = torch.randint(-7, 8, (rows, Fin), generator=g).float()
    W = torch.randint(-5, 6, (Fout, Fin), generator=g).float()
    W[::3] += 0.5  # asymmetric, exercises the mid piece (x.5 needs more than 8 bits once |value| >= 128 only, so still exact)
    b = torch.randint(-9, 10, (Fout,), generator=g).float()
    ref = x @ W.t() + b
    y = hip.linear_fwd(x.to(DEV), W.to(DEV), b.to(DEV), None)
    assert torch.equal(y.cpu(), ref)
    # values that NEED all three pieces (24 significant bits): x = 1 + 2^-10 + 2^-20 etc. times small integers
    xs = x * (1 + 2.0 ** -10 + 2.0 ** -20)
    ref64 = xs.double() @ W.double().t() + b.double()
    y = hip.linear_fwd(xs.to(DEV), W.to(DEV), b.to(DEV), None)
    assert float((y.cpu().double() - ref64).abs().max()) <= 2.0 ** -21 * float(ref64.abs().max())


@pytest.mark.parametrize("rows,Fin,Fout,scale", [(8192, 64, 64, 1.0), (8192, 64, 64, 1e-3), (5000, 48, 32, 30.0)])
def test_x3_linear_fwd_is_as_accurate_as_fp32(hip, rows, Fin, Fout, scale):
    """fp64 arbitration of the split-operand forward: its distance to the exact product must not exceed the
    distance of a plain fp32 GEMM (torch CPU) by more than rounding noise - 'fp32 accuracy', not 'bf16 accuracy'."""
    x, W, b = rnd(rows, Fin, seed=11, scale=scale), rnd(Fout, Fin, seed=12, scale=0.3), rnd(Fout, seed=13)
    a = torch.tensor([0.25])
    ref64 = P.prelu(x, a).double() @ W.double().t() + b.double()
    ref32 = P.prelu(x, a) @ W.t() + b
    y = hip.linear_fwd(x.to(DEV), W.to(DEV), b.to(DEV), a.to(DEV)).cpu().double()
    e_hip, e_32 = float((y - ref64).norm()), float((ref32.double() - ref64).norm())
    m_hip, m_32 = float((y - ref64).abs().max()), float((ref32.double() - ref64).abs().max())
    print(f"x3 forward: fro err {e_hip:.3e} (fp32 GEMM {e_32:.3e}), max err {m_hip:.3e} (fp32 GEMM {m_32:.3e})")
    assert e_hip <= 1.5 * e_32 + 1e-7 * float(ref64.norm())
    assert m_hip <= 2.0 * m_32 + 1e-7 * float(ref64.abs().max())


@pytest.mark.parametrize("rows,Fin,Fout", [(1000, 72, 48), (4099, 64, 64), (300, 48, 33), (77, 44, 128), (2000, 128, 128)])
def test_linear_bwd(hip, rows, Fin, Fout):
    x = rnd(rows, Fin, seed=1).requires_grad_()
    W = rnd(Fout, Fin, seed=2, scale=0.2).requires_grad_()
    b = rnd(Fout, seed=3).requires_grad_()
    a = torch.tensor([0.25], requires_grad=True)
    dy = rnd(rows, Fout, seed=4)
    (P.prelu(x, a) @ W.t() + b).backward(dy)
    xd, Wd, ad, dyd = x.detach().to(DEV), W.detach().to(DEV), a.detach().to(DEV), dy.to(DEV)
    da = torch.zeros(1, device=DEV)
    dx = hip.linear_bwd_dx(dyd, Wd, xd, ad, da)
    dW, db = torch.empty(Fout, Fin, device=DEV), torch.empty(Fout, device=DEV)
    hip.linear_bwd_dw(dyd, xd, ad, dW, db, False)
    assert rel(dx, x.grad) < TOL and rel(dW, W.grad) < TOL and rel(db, b.grad) < TOL
    assert rel(da, a.grad) < 1e-4
    # accumulate flag adds on top
    hip.linear_bwd_dw(dyd, xd, ad, dW, db, True)
    assert rel(dW, 2 * W.grad) < TOL and rel(db, 2 * b.grad) < TOL
    # no activation on the input
    dx2 = hip.linear_bwd_dx(dyd, Wd, None, None, None)
    assert rel(dx2, dy @ W.detach()) < TOL


@pytest.mark.parametrize("rows,Fin,Fout", [(1000, 64, 64), (4099, 48, 64), (300, 48, 48), (129, 72, 48), (5000, 64, 48),
                                           (2000, 32, 32), (77, 48, 33), (700, 128, 128), (1, 64, 64)])
@pytest.mark.parametrize("with_slope", [True, False])
def test_linear_bwd_all(hip, rows, Fin, Fout, with_slope):
    """Fused dX/dW/db/slope/column-sum backward (and its fallback for unsupported shapes)."""
    x = rnd(rows, Fin, seed=1).requires_grad_()
    W = rnd(Fout, Fin, seed=2, scale=0.2).requires_grad_()
    b = rnd(Fout, seed=3).requires_grad_()
    a = torch.tensor([0.25], requires_grad=True)
    dy = rnd(rows, Fout, seed=4)
    ((P.prelu(x, a) if with_slope else x) @ W.t() + b).backward(dy)
    xd, Wd, dyd = x.detach().to(DEV), W.detach().to(DEV), dy.to(DEV)
    ad = a.detach().to(DEV) if with_slope else None
    da = torch.zeros(1, device=DEV) if with_slope else None
    dW, db, cs = torch.ones(Fout, Fin, device=DEV), torch.ones(Fout, device=DEV), torch.ones(Fin, device=DEV)
    dx = hip.linear_bwd_all(dyd, Wd, xd, ad, da, dW, db, cs, True)  # accumulate on top of ones
    assert rel(dx, x.grad) < TOL
    assert rel(dW - 1, W.grad) < 2e-5 and rel(db - 1, b.grad) < 2e-5
    assert rel(cs - 1, x.grad.sum(0)) < 1e-4
    if with_slope:
        assert rel(da, a.grad) < 1e-4
    dW2, db2 = torch.empty(Fout, Fin, device=DEV), torch.empty(Fout, device=DEV)
    hip.linear_bwd_all(dyd, Wd, xd, ad, torch.zeros(1, device=DEV) if with_slope else None, dW2, db2, None, False)
    assert rel(dW2, W.grad) < 2e-5 and rel(db2, b.grad) < 2e-5


@pytest.mark.parametrize("rows,Fin,Fout", [(4096, 64, 64), (1000, 48, 64), (777, 64, 33), (129, 36, 16), (63, 64, 48)])
def test_x3_linear_bwd_exact_on_integers(hip, rows, Fin, Fout):
    """Fused backward on the bf16 matrix pipe with 3-way operand splitting (csrc/linear_x3.hip): with small-integer
    data every product and partial sum is exact in fp32, so dX, dW, db, colsum and the slope gradient must be
    BIT-equal to the fp64 result - any slip in the row images, the transposing LDS reads (dW sums over rows), the
    W^T fragments or the partial records shows up as a wrong integer."""
    g = torch.Generator().manual_seed(7)
    x = torch.randint(-6, 7, (rows, Fin), generator=g).float()
    W = torch.randint(-4, 5, (Fout, Fin), generator=g).float()
    dy = torch.randint(-3, 4, (rows, Fout), generator=g).float()
    a = torch.tensor([0.5])
    xr, Wr, ar = x.double().requires_grad_(), W.double().requires_grad_(), a.double().requires_grad_()
    br = torch.zeros(Fout, dtype=torch.float64, requires_grad=True)
    (torch.where(xr > 0, xr, ar * xr) @ Wr.t() + br).backward(dy.double())
    ldy = (Fout + 3) // 4 * 4  # Fout = 33: rows of dy padded to 36 floats, junk (finite) in the padding
    dyd = torch.full((rows, ldy), 77.0, device=DEV)
    dyd[:, :Fout] = dy.to(DEV)
    dW, db, cs = torch.empty(Fout, Fin, device=DEV), torch.empty(Fout, device=DEV), torch.empty(Fin, device=DEV)
    da = torch.zeros(1, device=DEV)
    dx = hip.linear_bwd_all(dyd[:, :Fout], W.to(DEV), x.to(DEV), a.to(DEV), da, dW, db, cs, False)
    assert torch.equal(dx.cpu().double(), xr.grad)
    assert torch.equal(dW.cpu().double(), Wr.grad) and torch.equal(db.cpu().double(), br.grad)
    assert torch.equal(cs.cpu().double(), xr.grad.sum(0))
    assert float(da.cpu().double()) == float(ar.grad)


@pytest.mark.parametrize("rows,Fin,Fout", [(8192, 64, 64), (5000, 48, 33)])
def test_x3_linear_bwd_is_as_accurate_as_fp32(hip, rows, Fin, Fout):
    """fp64 arbitration of the split-operand backward against a plain fp32 autograd run (torch CPU)."""
    x, W, dy = rnd(rows, Fin, seed=21), rnd(Fout, Fin, seed=22, scale=0.3), rnd(rows, Fout, seed=23)
    a = torch.tensor([0.25])
    res = {}
    for dt in (torch.float64, torch.float32):
        xr, Wr, ar = x.to(dt).requires_grad_(), W.to(dt).requires_grad_(), a.to(dt).requires_grad_()
        br = torch.zeros(Fout, dtype=dt, requires_grad=True)
        (torch.where(xr > 0, xr, ar * xr) @ Wr.t() + br).backward(dy.to(dt))
        res[dt] = [t.grad.double() for t in (xr, Wr, br, ar)]
    ldy = (Fout + 3) // 4 * 4
    dyd = torch.zeros(rows, ldy, device=DEV)
    dyd[:, :Fout] = dy.to(DEV)
    dW, db = torch.empty(Fout, Fin, device=DEV), torch.empty(Fout, device=DEV)
    da = torch.zeros(1, device=DEV)
    dx = hip.linear_bwd_all(dyd[:, :Fout], W.to(DEV), x.to(DEV), a.to(DEV), da, dW, db, None, False)
    for name, got, r64, r32 in zip(("dx", "dW", "db", "d_slope"), (dx, dW, db, da), res[torch.float64], res[torch.float32]):
        e_hip, e_32 = float((got.cpu().double() - r64).norm()), float((r32 - r64).norm())
        print(f"x3 backward {name}: err {e_hip:.3e} (fp32 autograd {e_32:.3e}) of {float(r64.norm()):.3e}")
        # the slope gradient is ONE number summed over every negative element (here 200x cancellation): the bf16
        # pipe's accumulate carries a bias of ~0.05 ulp that a sum of this length exposes (5e-6 relative; 1e-5 allowed)
        assert e_hip <= 2.0 * e_32 + (1e-5 if name == "d_slope" else 1e-6) * float(r64.norm()), name


@pytest.mark.parametrize("rows,Fin,Fout", [(1000, 64, 64), (300, 48, 36), (129, 72, 48), (700, 128, 128)])
def test_linear_bwd_all_accumulate_bits(hip, rows, Fin, Fout):
    """dW, db and colsum_dx belong to different parameters: each has its own accumulate bit
    (GCL_ACC_DW / GCL_ACC_DB / GCL_ACC_COLSUM), on the fused kernel and on the fallback."""
    x, W, dy = rnd(rows, Fin, seed=1), rnd(Fout, Fin, seed=2, scale=0.2), rnd(rows, Fout, seed=4)
    a = torch.tensor([0.25])
    xr, Wr, ar = x.clone().requires_grad_(), W.clone().requires_grad_(), a.clone().requires_grad_()
    br = torch.zeros(Fout, requires_grad=True)
    (P.prelu(xr, ar) @ Wr.t() + br).backward(dy)
    xd, Wd, dyd, ad = x.to(DEV), W.to(DEV), dy.to(DEV), a.to(DEV)
    for bits in range(8):
        acc = [bool(bits & 1), bool(bits & 2), bool(bits & 4)]
        dW, db, cs = (torch.full((Fout, Fin), 3.0, device=DEV), torch.full((Fout,), 5.0, device=DEV),
                      torch.full((Fin,), 7.0, device=DEV))
        hip.linear_bwd_all(dyd, Wd, xd, ad, torch.zeros(1, device=DEV), dW, db, cs, acc[0], acc_db=acc[1],
                           acc_colsum=acc[2])
        assert rel(dW - (3.0 if acc[0] else 0.0), Wr.grad) < 2e-5, bits
        assert rel(db - (5.0 if acc[1] else 0.0), br.grad) < 2e-5, bits
        assert rel(cs - (7.0 if acc[2] else 0.0), xr.grad.sum(0)) < 1e-4, bits


def test_deferred_reductions_equal_immediate(hip):
    """gcl_linear_bwd_all_deferred + gcl_reduce_jobs: the final passes of several layers (different shapes, a shared
    slope gradient, one layer that runs twice = duplicate destinations, a shape that falls back to the separate
    kernels) run in one launch and give the same dW / db / colsum / slope as the immediate form."""
    shapes = [(3000, 64, 64), (777, 48, 33), (5000, 64, 48), (300, 128, 128), (3000, 64, 64)]
    a = torch.tensor([0.25], device=DEV)

    def run(deferred):
        outs, da = [], torch.zeros(1, device=DEV)
        if deferred:
            hip.defer_begin()
        keep = {}
        for k, (rows, Fin, Fout) in enumerate(shapes):
            x, W, dy = rnd(rows, Fin, seed=10 + k).to(DEV), rnd(Fout, Fin, seed=20 + k, scale=0.2).to(DEV), rnd(rows, Fout, seed=30 + k).to(DEV)
            ldy = (Fout + 3) // 4 * 4
            dyp = torch.zeros(rows, ldy, device=DEV)
            dyp[:, :Fout] = dy
            if k == 4:  # the same layer again: accumulates on top of call 0's destinations
                dW, db, cs = keep[0]
                acc = True
            else:
                dW, db, cs = torch.full((Fout, Fin), 2.0, device=DEV), torch.full((Fout,), 3.0, device=DEV), torch.full((Fin,), 4.0, device=DEV)
                keep[k] = (dW, db, cs)
                acc = k % 2 == 1
            dx = hip.linear_bwd_all(dyp[:, :Fout], W, x, a, da, dW, db, cs, acc, acc_db=not acc, acc_colsum=acc)
            outs.append(dx)
        if deferred:
            hip.defer_flush()
        for k in sorted(keep):
            outs.extend(keep[k])
        outs.append(da)
        return outs

    imm, dfr = run(False), run(True)
    assert not hip._deferred.active and not hip._deferred.jobs
    for u, v in zip(imm, dfr):
        assert rel(v, u) < 2e-6


def test_reduce_jobs_more_than_one_launch_with_an_empty_job(hip):
    """gcl_reduce_jobs takes 16 jobs per launch and skips empty ones (`nparts == 0`: that call reduced on the spot,
    legal input per include/gcl.h).  With 19 jobs and an empty one among the first 16, the jobs behind it must be
    reduced exactly ONCE (restarting the scan at base + 16 used to reduce job 17 twice: doubled dW with the
    accumulate bit, doubled slope gradient always)."""
    import ctypes as C

    n_layers, rows, F = 18, 1500, 64
    a = torch.tensor([0.25], device=DEV)
    xs = [rnd(rows, F, seed=100 + k).to(DEV) for k in range(n_layers)]
    Ws = [rnd(F, F, seed=200 + k, scale=0.2).to(DEV) for k in range(n_layers)]
    dys = [rnd(rows, F, seed=300 + k).to(DEV) for k in range(n_layers)]

    def run(deferred):
        dWs = [torch.full((F, F), 1.0, device=DEV) for _ in range(n_layers)]
        dbs = [torch.full((F,), 1.0, device=DEV) for _ in range(n_layers)]
        da = torch.zeros(1, device=DEV)
        if deferred:
            hip.defer_begin()
        for k in range(n_layers):
            hip.linear_bwd_all(dys[k], Ws[k], xs[k], a, da, dWs[k], dbs[k], None, True)  # accumulate onto the 1.0s
        if deferred:
            d = hip._deferred
            jobs = list(d.jobs)
            assert len(jobs) == n_layers
            jobs.insert(3, hip.ReduceJob())  # an empty job inside the first launch's window
            arr = (hip.ReduceJob * len(jobs))(*jobs)
            hip._check(hip.lib().gcl_reduce_jobs(C.cast(arr, C.c_void_p), len(jobs), hip._stream()))
            torch.cuda.synchronize()
            hip.defer_flush(drop=True)
        return dWs, dbs, da

    (w0, b0, a0), (w1, b1, a1) = run(False), run(True)
    for k in range(n_layers):
        assert rel(w1[k], w0[k]) < 2e-6 and rel(b1[k], b0[k]) < 2e-6, k
    assert rel(a1, a0) < 2e-6


def test_linear_mfma_equals_valu(hip, monkeypatch):
    """The fp32 MFMA path and the plain VALU path of the same entry point agree (both fp32 FMA chains)."""
    import os
    import subprocess
    import sys

    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import torch\n"
        "from graphcast_lite_amd import hip\n"
        "g = torch.Generator().manual_seed(5)\n"
        "x = torch.randn(777, 72, generator=g).cuda(); W = (torch.randn(48, 72, generator=g) * 0.2).cuda()\n"
        "y = hip.linear_fwd(x, W, None, None)\n"
        "torch.save(y.cpu(), sys.argv[1])\n" % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    outs = []
    for impl in ("mfma", "valu"):
        path = f"/tmp/gcl_lin_{impl}.pt"
        env = dict(os.environ, GCL_LINEAR_IMPL=impl)
        subprocess.run([sys.executable, "-c", code, path], check=True, env=env)
        outs.append(torch.load(path))
    assert rel(outs[0], outs[1]) < 2e-6


@pytest.mark.parametrize("levels,F,B", [([1, 2], 64, 3), ([1, 2], 33, 2), ([3, 5], 64, 9), ([1, 2], 128, 8),
                                        ([0], 19, 1), ([1, 2], 48, 16), ([1, 2], 4, 2), ([1, 2], 132, 1)])
def test_gcn_aggregate_mesh(hip, levels, F, B):
    g = build_graphs(experiment("baseline", mesh_levels=levels))
    n = g["M"]
    h, b = rnd(B, n, F, seed=1), rnd(F, seed=2)
    ei, w = P.gcn_norm(g["proc"], n, torch.float32)
    ref = P._propagate_sum(h, ei, w, n) + b
    G = hip.Graph(g["proc"], n, hip.GRAPH_GCN)
    assert G.e == g["proc"].shape[1] + n
    y = hip.aggregate(G, h.to(DEV), b.to(DEV))
    assert rel(y, ref) < TOL
    # transpose = gradient of the aggregation
    dy = rnd(B, n, F, seed=3)
    hr = h.clone().requires_grad_()
    (P._propagate_sum(hr, ei, w, n)).backward(dy)
    dh = hip.aggregate(G, dy.to(DEV), None, transpose=True)
    assert rel(dh, hr.grad) < TOL


def _tiled(g, leaf=64):
    """The mesh graph with nodes renamed to tile order (what models.WeatherPrediction hands the processor)."""
    from graphcast_lite_amd.mesh import tile_order

    n = g["M"]
    deg = torch.bincount(g["proc"][1], minlength=n).numpy()
    order = torch.from_numpy(np.ascontiguousarray(tile_order(g["mesh"].vertices, leaf, degree=deg)))
    pos = torch.empty(n, dtype=torch.int64)
    pos[order] = torch.arange(n)
    return pos[g["proc"]], order, pos


@pytest.mark.parametrize("levels,F,B", [([3, 5], 64, 9), ([3, 5], 64, 64), ([3, 5], 48, 3), ([2, 4], 64, 17), ([3, 5], 128, 2),
                                        ([2, 3], 64, 8)])
def test_gcn_aggregate_source_tiles(hip, levels, F, B, monkeypatch):
    """Mesh rows in tile order: gcl_aggregate stages a tile's distinct source rows once in LDS (agg_halo_kernel,
    include/gcl.h: gcl_graph_halo_info).  Against the oracle on the renamed graph, un-renamed against the oracle
    on the reference graph (renaming must be invisible), and BIT-equal to the per-edge gather kernel, forward and
    transposed, with and without bias."""
    g = build_graphs(experiment("baseline", mesh_levels=levels))
    n = g["M"]
    ei_t, order, pos = _tiled(g)
    G = hip.Graph(ei_t, n, hip.GRAPH_GCN)
    assert G.halo_info(False, 64) is not None and G.halo_info(True, 64) is not None, "tile order must enable the source-tile layout"
    h, b = rnd(B, n, F, seed=1), rnd(F, seed=2)
    e_ref, w_ref = P.gcn_norm(g["proc"], n, torch.float32)
    ref = P._propagate_sum(h, e_ref, w_ref, n) + b          # reference numbering
    for tr, bias in ((False, b), (True, None)):
        outs = {}
        for mode in ("1", "0"):
            monkeypatch.setenv("GCL_AGG_HALO", mode)
            outs[mode] = hip.aggregate(G, h[:, order].contiguous().to(DEV), None if bias is None else bias.to(DEV), transpose=tr).cpu()
        assert torch.equal(outs["1"], outs["0"]), "source-tile kernel and per-edge kernel differ in some bit"
        if not tr:
            assert rel(outs["1"][:, pos], ref) < TOL
        else:
            hr = h.clone().requires_grad_()
            P._propagate_sum(hr, e_ref, w_ref, n).backward(h)
            assert rel(outs["1"][:, pos], hr.grad) < TOL


def test_gcn_aggregate_source_tiles_long_rows(hip, monkeypatch):
    """Rows beyond the 16 edge records a lane group holds (finished from the CSR arrays), heavy rows (> 64 edges: their
    own kernel), a ragged last tile and non-finite inputs next to padded slots: a ring lattice (6 neighbours) with a
    few long rows.  Bit-equal to the per-edge kernel and within tolerance of the oracle."""
    rng = np.random.default_rng(7)
    n = 64 * 37 + 19
    idx = np.arange(n)
    src = np.concatenate([(idx + d) % n for d in (-3, -2, -1, 1, 2, 3)])
    dst = np.tile(idx, 6)
    extra_s, extra_d = [], []
    for r, d in ((100, 20), (101, 40), (777, 17), (1500, 90), (n - 1, 30)):
        nb = (r + rng.choice(np.arange(4, 200), size=d - 7, replace=False) * rng.choice([-1, 1], size=d - 7)) % n
        extra_s.append(nb); extra_d.append(np.full(nb.size, r))
    ei = torch.from_numpy(np.stack([np.concatenate([src] + extra_s), np.concatenate([dst] + extra_d)]).astype(np.int64))
    G = hip.Graph(ei, n, hip.GRAPH_GCN)
    assert G.halo_info(False, 64) is not None and G.max_in_degree > 64
    B, F = 5, 64
    h = rnd(B, n, F, seed=4)
    h[0, 9, 3] = float("inf")  # must reach only the rows that really read row 9
    e2, w = P.gcn_norm(ei, n, torch.float32)
    ref = P._propagate_sum(h, e2, w, n)
    for tr in (False, True):
        outs = {}
        for mode in ("1", "0"):
            monkeypatch.setenv("GCL_AGG_HALO", mode)
            outs[mode] = hip.aggregate(G, h.to(DEV), None, transpose=tr).cpu()
        assert torch.equal(outs["1"].view(torch.int32), outs["0"].view(torch.int32))
    fin = torch.isfinite(ref)
    monkeypatch.setenv("GCL_AGG_HALO", "1")
    y = hip.aggregate(G, h.to(DEV), None).cpu()
    assert torch.equal(torch.isfinite(y), fin)
    assert rel(torch.where(fin, y, torch.zeros(())), torch.where(fin, ref, torch.zeros(()))) < TOL


def test_gcn_aggregate_bipartite_and_padded_ld(hip):
    """Encoder (skewed in-degree, most rows self-loop only) and decoder graphs, padded scratch input."""
    g = build_graphs(experiment("baseline", mesh_levels=[3, 5]))
    n = g["G"] + g["M"]
    for ei_ref in (g["enc"], g["dec"]):
        G = hip.Graph(ei_ref, n, hip.GRAPH_GCN)
        for F, ld in ((33, 36), (48, 48), (64, 64)):
            B = 2
            buf = rnd(B, n, ld, seed=F)
            h = buf[..., :F]
            ei, w = P.gcn_norm(ei_ref, n, torch.float32)
            ref = P._propagate_sum(h, ei, w, n)
            hd = buf.to(DEV)[..., :F]
            y = hip.aggregate(G, hd, None)
            assert rel(y, ref) < TOL


@pytest.mark.parametrize("F,B", [(64, 3), (128, 2), (33, 1)])
def test_aggregate_heavy_rows(hip, F, B):
    """Rows with hundreds of in-edges (polar mesh nodes of E_G2M at 512x256 reach 688) take the
    one-block-per-row path; forward and transpose, GCN and mean weights."""
    rng = np.random.default_rng(3)
    n = 2000
    hubs = np.array([5, 700, 1999])
    src, dst = [], []
    for h_, d in zip(hubs, (300, 65, 943)):
        s_ = rng.choice(n, size=d, replace=False)
        src.append(s_[s_ != h_]); dst.append(np.full((s_ != h_).sum(), h_))
    src.append(rng.integers(0, n, 3000)); dst.append(rng.integers(0, n, 3000))
    ei = torch.from_numpy(np.stack([np.concatenate(src), np.concatenate(dst)]).astype(np.int64))
    h = rnd(B, n, F, seed=1)
    for kind, ref_fn in ((hip.GRAPH_GCN, None), (hip.GRAPH_MEAN, None)):
        G = hip.Graph(ei, n, kind)
        assert G.max_in_degree > 64
        if kind == hip.GRAPH_GCN:
            e2, w = P.gcn_norm(ei, n, torch.float32)
            ref = P._propagate_sum(h, e2, w, n)
            hr = h.clone().requires_grad_()
            P._propagate_sum(hr, e2, w, n).backward(h)
        else:
            ref = P.simple_conv_mean(h, ei)
            hr = h.clone().requires_grad_()
            P.simple_conv_mean(hr, ei).backward(h)
        y = hip.aggregate(G, h.to(DEV), None)
        assert rel(y, ref) < TOL
        dh = hip.aggregate(G, h.to(DEV), None, transpose=True)
        assert rel(dh, hr.grad) < TOL


def test_mean_aggregate(hip):
    g = build_graphs(experiment("baseline", mesh_levels=[1, 2]))
    n = g["G"] + g["M"]
    x = rnd(2, n, 64, seed=4)
    for ei in (g["enc"], g["dec"]):
        G = hip.Graph(ei, n, hip.GRAPH_MEAN)
        y = hip.aggregate(G, x.to(DEV), None)
        ref = P.simple_conv_mean(x, ei)
        assert rel(y, ref) < TOL
        assert (y.cpu()[ref == 0] == 0).all()  # nodes without in-edges are exactly zero


@pytest.mark.parametrize("rows,F", [(1000, 64), (333, 128), (50, 33), (7, 12), (4096, 48)])
def test_layernorm(hip, rows, F):
    x = rnd(rows, F, seed=1).requires_grad_()
    gm = (torch.rand(F, generator=torch.Generator().manual_seed(2)) + 0.5).requires_grad_()
    bt = rnd(F, seed=3).requires_grad_()
    dy = rnd(rows, F, seed=4)
    ref = P.pyg_layer_norm(x, gm, bt, "node")
    ref.backward(dy)
    y, stats = hip.layernorm_fwd(x.detach().to(DEV), gm.detach().to(DEV), bt.detach().to(DEV))
    assert rel(y, ref) < TOL
    dg, db = torch.empty(F, device=DEV), torch.empty(F, device=DEV)
    dx = hip.layernorm_bwd(dy.to(DEV), x.detach().to(DEV), gm.detach().to(DEV), stats, dg, db, False)
    assert rel(dx, x.grad) < 2e-5 and rel(dg, gm.grad) < 2e-5 and rel(db, bt.grad) < TOL
    # the same pass can also sum dx over the rows (bias gradient of the layer below), with its own accumulate bit
    dg2, db2, cs = torch.ones(F, device=DEV), torch.ones(F, device=DEV), torch.full((F,), 2.0, device=DEV)
    dx2 = hip.layernorm_bwd(dy.to(DEV), x.detach().to(DEV), gm.detach().to(DEV), stats, dg2, db2, False, colsum_dx=cs,
                            acc_colsum=True)
    assert rel(dx2, dx) < 1e-6 and rel(dg2, dg) < 1e-6 and rel(db2, db) < 1e-6
    ref_cs = x.grad.double().sum(0)
    assert float((cs.cpu().double() - 2.0 - ref_cs).abs().max()) <= 1e-5 * float(x.grad.abs().sum(0).max()) + 1e-6
    hip.layernorm_bwd(dy.to(DEV), x.detach().to(DEV), gm.detach().to(DEV), stats, dg2, db2, True, colsum_dx=cs, acc_colsum=False)
    assert rel(dg2, 2 * gm.grad) < 2e-5 and float((cs.cpu().double() - ref_cs).abs().max()) <= 1e-5 * float(x.grad.abs().sum(0).max()) + 1e-6


@pytest.mark.parametrize("B,n,F", [(3, 500, 64), (1, 162, 33), (2, 3000, 128), (2, 7, 12)])
def test_graphnorm(hip, B, n, F):
    """PyG LayerNorm(mode="graph"): per-sample statistics over all n*F elements, eps added to the std."""
    x = (rnd(B, n, F, seed=1) * 2 + 0.5).requires_grad_()
    gm = (torch.rand(F, generator=torch.Generator().manual_seed(2)) + 0.5).requires_grad_()
    bt = rnd(F, seed=3).requires_grad_()
    dy = rnd(B, n, F, seed=4)
    ref = P.pyg_layer_norm(x, gm, bt, "graph")
    ref.backward(dy)
    y, stats = hip.graphnorm_fwd(x.detach().to(DEV), gm.detach().to(DEV), bt.detach().to(DEV))
    assert rel(y, ref) < TOL
    dg, db = torch.empty(F, device=DEV), torch.empty(F, device=DEV)
    dx = hip.graphnorm_bwd(dy.to(DEV), x.detach().to(DEV), gm.detach().to(DEV), stats, dg, db, False)
    assert rel(dx, x.grad) < 2e-5 and rel(dg, gm.grad) < 2e-5 and rel(db, bt.grad) < TOL
    const = hip.graphnorm_fwd(torch.full((1, 5, F), 2.5, device=DEV), gm.detach().to(DEV), bt.detach().to(DEV))[0]
    assert rel(const[0], bt.detach().expand(5, F)) < 1e-6  # 0 / (0 + eps)


@pytest.mark.parametrize("B,n,F,ldo", [(3, 500, 64, 64), (2, 77, 33, 36), (4, 1000, 128, 128)])
def test_layernorm_fwd_through_row_map(hip, B, n, F, ldo):
    """gcl_layernorm_fwd_map: row (b, i) is written to out[b, pos[i]] (pos[i] >= 0) and nowhere otherwise; statistics
    for every row.  Bit-equal to the dense LayerNorm followed by the gather; untouched rows keep their content."""
    gen = torch.Generator().manual_seed(3)
    x, gm, bt = rnd(B * n, F, seed=1), rnd(F, seed=2), rnd(F, seed=3)
    keep = torch.rand(n, generator=gen) < 0.45
    nk = int(keep.sum())
    head = 11
    pos = torch.full((n,), -1, dtype=torch.int32)
    pos[keep] = (head + torch.randperm(nk, generator=gen)).to(torch.int32)
    y_ref, st_ref = hip.layernorm_fwd(x.to(DEV), gm.to(DEV), bt.to(DEV), 1e-5)
    out = torch.full((B, head + nk, ldo), 7.0, device=DEV)
    st = hip.layernorm_fwd_map(x.to(DEV), gm.to(DEV), bt.to(DEV), 1e-5, out, pos.to(DEV))
    # (the two instantiations are separate compilations of the same expressions: equal to rounding, and bit-equal for the
    # 16-byte-row widths the model uses - the folded-vs-separate model test relies on that)
    y3 = y_ref.view(B, n, F)
    if F % 4 == 0:
        assert torch.equal(st, st_ref)
        assert torch.equal(out[:, pos[keep].long(), :F], y3[:, keep])
    assert rel(st, st_ref) < 1e-6 and rel(out[:, pos[keep].long(), :F], y3[:, keep]) < 1e-6
    assert (out[:, :head] == 7.0).all() and (out[..., F:] == 7.0).all()


def test_gather2_rows_batch_sum_dealt_to_samples(hip):
    """gcl_gather2_rows with sum_batch = R > 1: row i of the nd batch-summed rows is stored at dst[i / R, i % R] (the
    gradient of the shared rows of the compact pipeline lands R to a sample, straight in a row range of a larger buffer)."""
    B, n, F, R = 5, 300, 64, 7
    gen = torch.Generator().manual_seed(2)
    a = rnd(B, n, F, seed=1)
    m = torch.randint(-1, n, (B * R,), generator=gen).to(torch.int32)
    ref = torch.where((m >= 0)[:, None], a.sum(0)[m.clamp(min=0).long()], torch.zeros(1, F))
    flat = hip.gather2_rows(a.to(DEV), m.to(DEV), None, None, B * R, B, sum_batch=True)
    assert rel(flat[0], ref) < 1e-6
    buf = torch.full((B, 4 + R, F), 3.0, device=DEV)
    hip.gather2_rows(a.to(DEV), m.to(DEV), None, None, B * R, B, sum_batch=True, out=buf[:, 4:], deal=R)
    assert torch.equal(buf[:, 4:].reshape(B * R, F), flat[0]) and (buf[:, :4] == 3.0).all()


@pytest.mark.parametrize("rows,F", [(1000, 64), (13, 33), (5000, 128)])
def test_colsum(hip, rows, F):
    x = rnd(rows, F, seed=1)
    out = torch.ones(F, device=DEV)
    hip.colsum(x.to(DEV), out, True)
    assert rel(out, x.sum(0) + 1) < TOL


@pytest.mark.parametrize("levels,H,C,B", [([1, 2], 1, 64, 3), ([3, 5], 1, 64, 2), ([3, 5], 1, 64, 9), ([1, 2], 4, 64, 2), ([0], 2, 16, 1),
                                          ([1, 2], 1, 128, 2),
                                          # head counts the reference reports (README.md:148-150): 8 and 33 heads at C = 64
                                          # run as chunks of <= 4 heads; 3 and 6 heads exercise the uneven chunking
                                          ([1, 2], 8, 64, 2), ([1, 2], 33, 64, 2), ([3, 5], 8, 64, 9), ([0], 3, 64, 1),
                                          ([1, 2], 6, 32, 2),
                                          # head widths whose C / 4 is not a power of two (the reference's GATConv takes any
                                          # hidden width, src/models.py:332-358; 96 is the wb2_64x32_15f stack width): a head
                                          # is padded to the next power-of-two lane count, its idle lanes contribute zeros
                                          ([1, 2], 1, 48, 3), ([1, 2], 2, 96, 2), ([1, 2], 3, 48, 2), ([3, 5], 1, 96, 9),
                                          ([0], 5, 20, 1), ([1, 2], 1, 12, 2), ([1, 2], 4, 36, 2)])
def test_gat_fwd_bwd(hip, levels, H, C, B):
    g = build_graphs(experiment("baseline", mesh_levels=levels))
    n = g["M"]
    Fin = 24
    x = rnd(B, n, Fin, seed=1).requires_grad_()
    W = rnd(H * C, Fin, seed=2, scale=0.3).requires_grad_()
    a_s, a_d = rnd(1, H, C, seed=3, scale=0.3).requires_grad_(), rnd(1, H, C, seed=4, scale=0.3).requires_grad_()
    b = rnd(C, seed=5).requires_grad_()
    y_ref, ei2, alpha_ref = P.gat_conv(x, g["proc"], W, a_s, a_d, b, H)
    dy = rnd(B, n, C, seed=6)
    y_ref.backward(dy)
    h_ref = (x.detach() @ W.detach().t())

    G = hip.Graph(g["proc"], n, hip.GRAPH_GAT)
    assert torch.equal(G.export_edges(), ei2)
    hd = h_ref.to(DEV)
    y, s_src, s_dst, alpha = hip.gat_fwd(G, hd, a_s.detach().reshape(-1).to(DEV), a_d.detach().reshape(-1).to(DEV),
                                         b.detach().to(DEV), H, C)
    assert rel(y, y_ref) < TOL
    al_e = torch.stack([hip.gat_alpha_edge_order(G, alpha[i], H) for i in range(B)])
    assert rel(al_e, alpha_ref) < TOL

    d_as, d_ad, d_b = torch.empty(H * C, device=DEV), torch.empty(H * C, device=DEV), torch.empty(C, device=DEV)
    dh = hip.gat_bwd(G, dy.to(DEV), hd, a_s.detach().reshape(-1).to(DEV), a_d.detach().reshape(-1).to(DEV), s_src,
                     s_dst, alpha, d_as, d_ad, d_b, False, H, C)
    # dh -> dW, dx through the dense transform (checked separately); compare dh via dW = dh^T x
    # (the products that turn dh into dW / dx are formed in float64 here, so the check measures dh, not this matmul)
    dW = dh.reshape(-1, H * C).t().cpu().double() @ x.detach().reshape(-1, Fin).double()
    assert rel(dW, W.grad) < 5e-5
    # leaky_relu has a kink at 0: an edge whose score lands within rounding of it takes slope 1 on one side of the
    # comparison and 0.2 on the other (seen on the [3,5] mesh: one edge of 82k x 9 x 8) - those edges' two end rows
    # are left out of the ELEMENTWISE dx check (they stay in dW above, where one edge is far below the tolerance)
    h64 = h_ref.double().reshape(B, n, H, C)
    e = ((h64 * a_s.detach().double()).sum(-1)[:, ei2[0]] + (h64 * a_d.detach().double()).sum(-1)[:, ei2[1]]).abs()
    near = (e < 2e-6).any(dim=2).any(dim=0)  # fp32 rounding of a score of size O(1)
    keep = torch.ones(n, dtype=torch.bool)
    keep[ei2[0][near]] = False
    keep[ei2[1][near]] = False
    assert int((~keep).sum()) <= max(8, n // 200)
    dx = dh.cpu().double() @ W.detach().double()
    assert rel(dx[:, keep], x.grad[:, keep]) < 5e-5
    assert float((dx - x.grad).norm() / x.grad.norm()) < 5e-5
    assert rel(d_as.cpu(), a_s.grad.reshape(-1)) < 5e-5 and rel(d_ad.cpu(), a_d.grad.reshape(-1)) < 5e-5
    assert rel(d_b, b.grad) < TOL


@pytest.mark.parametrize("levels,C,B", [([3, 5], 64, 9), ([3, 5], 64, 64), ([2, 4], 128, 3), ([2, 3], 64, 2)])
def test_gat_fwd_source_tiles(hip, levels, C, B, monkeypatch):
    """One head on mesh rows in tile order: gcl_gat_fwd does the whole layer from one LDS image per tile
    (gat_halo_fwd_kernel).  Output, attention weights (edge order) and the saved scores against the oracle's GATConv on
    the REFERENCE numbering (renaming must be invisible), and against the per-edge kernels on the same graph; the
    backward then runs on what the forward saved."""
    g = build_graphs(experiment("baseline", mesh_levels=levels))
    n, H, Fin = g["M"], 1, 24
    ei_t, order, pos = _tiled(g)
    nb = min(B, 3)
    x = rnd(B, n, Fin, seed=1)
    W = rnd(C, Fin, seed=2, scale=0.3)
    a_s, a_d, b = rnd(1, 1, C, seed=3, scale=0.3), rnd(1, 1, C, seed=4, scale=0.3), rnd(C, seed=5)
    xr = x[:nb].clone().requires_grad_()
    y_ref, ei2, alpha_ref = P.gat_conv(xr, g["proc"], W, a_s, a_d, b, H)
    dy = rnd(B, n, C, seed=6)
    y_ref.backward(dy[:nb])
    G = hip.Graph(ei_t, n, hip.GRAPH_GAT)
    assert G.halo_info(False, 64) is not None
    hd = (x @ W.t())[:, order].contiguous().to(DEV)
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("GCL_GAT_HALO", mode)
        outs[mode] = hip.gat_fwd(G, hd, a_s.reshape(-1).to(DEV), a_d.reshape(-1).to(DEV), b.to(DEV), H, C)
    y, s_src, s_dst, alpha = outs["1"]
    for a_, b_ in zip(outs["1"], outs["0"]):
        assert rel(a_, b_) < 2e-6
    assert rel(y[:nb, pos], y_ref) < TOL
    # attention weights in PyG edge order: edge e of the reference list is edge e of the renamed list (same order); the
    # self-loops PyG appends are numbered by node, so theirs follow the renaming
    al_e = torch.stack([hip.gat_alpha_edge_order(G, alpha[i], H) for i in range(nb)]).cpu()
    E = g["proc"].shape[1]
    assert rel(al_e[:, :E], alpha_ref[:, :E]) < TOL and rel(al_e[:, E:][:, pos], alpha_ref[:, E:]) < TOL
    d_as, d_ad, d_b = torch.empty(C, device=DEV), torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    dh = hip.gat_bwd(G, dy[:, order].contiguous().to(DEV), hd, a_s.reshape(-1).to(DEV), a_d.reshape(-1).to(DEV), s_src, s_dst,
                     alpha, d_as, d_ad, d_b, False, H, C)
    dx = dh[:nb, pos].cpu().double() @ W.double()
    assert float((dx - xr.grad).norm() / xr.grad.norm()) < 5e-5
    # the per-edge backward kernels on the same inputs: same dh and attention-vector gradients
    monkeypatch.setenv("GCL_GAT_HALO", "0")
    e_as, e_ad, e_b = torch.empty(C, device=DEV), torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    dh0 = hip.gat_bwd(G, dy[:, order].contiguous().to(DEV), hd, a_s.reshape(-1).to(DEV), a_d.reshape(-1).to(DEV), s_src, s_dst,
                      alpha, e_as, e_ad, e_b, False, H, C)
    assert float((dh - dh0).norm() / dh0.norm()) < 2e-6
    assert rel(d_as, e_as) < 2e-5 and rel(d_ad, e_ad) < 2e-5 and rel(d_b, e_b) < 1e-6
    if B == nb:  # the oracle saw the whole batch: its parameter gradients are comparable
        a_sr, a_dr = a_s.clone().requires_grad_(), a_d.clone().requires_grad_()
        yr, _, _ = P.gat_conv(x, g["proc"], W, a_sr, a_dr, b, H)
        yr.backward(dy)
        assert rel(d_as.cpu(), a_sr.grad.reshape(-1)) < 5e-5 and rel(d_ad.cpu(), a_dr.grad.reshape(-1)) < 5e-5


@pytest.mark.parametrize("levels,C,B", [([3, 5], 64, 9), ([2, 4], 128, 3)])
def test_gat_through_row_table(hip, levels, C, B):
    """gcl_gat_fwd_tab / gcl_gat_bwd_tab: the rows of h are read through a row table (own rows of a compact tensor or
    batch-invariant flat rows).  Everything must be BIT-equal to the same calls on the materialised [B, n, C] rows."""
    g = build_graphs(experiment("baseline", mesh_levels=levels))
    n = g["M"]
    ei_t, order, pos = _tiled(g)
    G = hip.Graph(ei_t, n, hip.GRAPH_GAT)
    assert hip.gat_tab_ok(G, 1, C)
    nc = n // 3 + 11
    gen = torch.Generator().manual_seed(7)
    hc = rnd(B, nc, C, seed=1)
    own = torch.rand(n, generator=gen) < 0.2
    tab = torch.where(own, torch.randint(0, nc, (n,), generator=gen), -torch.randint(0, B * nc, (n,), generator=gen) - 1).to(torch.int32)
    h = torch.where(own[None, :, None], hc[:, tab.clamp(min=0).long()], hc.reshape(B * nc, C)[(-tab.long() - 1).clamp(min=0)][None]).contiguous()
    a_s, a_d, b = rnd(C, seed=3, scale=0.3).to(DEV), rnd(C, seed=4, scale=0.3).to(DEV), rnd(C, seed=5).to(DEV)
    ref = hip.gat_fwd(G, h.to(DEV), a_s, a_d, b, 1, C)
    got = hip.gat_fwd(G, hc.to(DEV), a_s, a_d, b, 1, C, tab=tab.to(DEV))
    for r_, g_ in zip(ref, got):
        assert torch.equal(r_, g_)
    dy = rnd(B, n, C, seed=6).to(DEV)
    outs = []
    for hh, tt in ((h.to(DEV), None), (hc.to(DEV), tab.to(DEV))):
        d_as, d_ad, d_b = torch.empty(C, device=DEV), torch.empty(C, device=DEV), torch.empty(C, device=DEV)
        dh = hip.gat_bwd(G, dy, hh, a_s, a_d, ref[1], ref[2], ref[3], d_as, d_ad, d_b, False, 1, C, tab=tt)
        outs.append((dh, d_as, d_ad, d_b))
    for r_, g_ in zip(outs[0], outs[1]):
        assert torch.equal(r_, g_)


def test_gat_unsupported_geometry_is_reported(hip):
    g = build_graphs(experiment("baseline", mesh_levels=[0]))
    G = hip.Graph(g["proc"], 12, hip.GRAPH_GAT)
    h = torch.zeros(1, 12, 260, device=DEV)
    z = torch.zeros(260, device=DEV)
    with pytest.raises(RuntimeError, match="unsupported head width"):  # one head must fit the 64 lanes x 4 channels of a wave
        hip.gat_fwd(G, h, z, z, torch.zeros(260, device=DEV), 1, 260)
    with pytest.raises(RuntimeError, match="multiple of 4"):
        hip.gat_fwd(G, torch.zeros(1, 12, 36, device=DEV), torch.zeros(34, device=DEV), torch.zeros(34, device=DEV),
                    torch.zeros(34, device=DEV), 1, 34)


def test_gat_prune(hip):
    g = build_graphs(experiment("baseline", mesh_levels=[1, 2]))
    n = g["M"]
    G = hip.Graph(g["proc"], n, hip.GRAPH_GAT)
    alpha = torch.rand(G.e, generator=torch.Generator().manual_seed(3))
    ei = G.export_edges()
    for thr in (0.0, 0.3, 0.999, 2.0):
        kept = hip.gat_prune(G, alpha.to(DEV), thr)
        ref, _ = P.sparse_gat_prune(ei, alpha, thr)
        assert torch.equal(kept, ref)
    # a graph rebuilt from its own export keeps the tensor identity (no CSR rebuild per step)
    ei_dev = ei.to(DEV)
    G2 = hip.Graph(ei_dev, n, hip.GRAPH_GAT)
    assert G2.edges_with_loops(ei_dev.device) is ei_dev


def test_assemble_input(hip):
    B, G, M, Cd, Cs = 3, 50, 20, 66, 6
    x, gs, ms = rnd(B, G, Cd, seed=1), rnd(G, Cs, seed=2), rnd(M, Cs, seed=3)
    out = hip.assemble_input(x.to(DEV), gs.to(DEV), ms.to(DEV)).cpu()
    ref = torch.cat([torch.cat([x, gs.expand(B, G, Cs)], -1), torch.cat([torch.zeros(B, M, Cd), ms.expand(B, M, Cs)], -1)], 1)
    assert torch.equal(out, ref)


def test_weighted_mse(hip):
    import os

    from conftest import GOLDEN
    from graphcast_lite_amd import train as TR

    v = np.load(os.path.join(GOLDEN, "loss_vectors.npz"))
    pred, target = torch.from_numpy(v["pred"]).to(DEV), torch.from_numpy(v["target"]).to(DEV)
    lat, chan, sm = (torch.from_numpy(v[k]).to(DEV) for k in ("lat_w", "chan_mask", "spatial_mask"))
    assert TR.weighted_mse_loss(pred, target).item() == pytest.approx(float(v["loss_plain"]), rel=1e-5)
    assert TR.weighted_mse_loss(pred, target, lat).item() == pytest.approx(float(v["loss_lat"]), rel=1e-5)
    assert TR.weighted_mse_loss(pred, target, lat, chan).item() == pytest.approx(float(v["loss_lat_chan"]), rel=1e-5)
    assert TR.weighted_mse_loss(pred, target, lat, chan, sm).item() == pytest.approx(float(v["loss_all"]), rel=1e-5)
    np.testing.assert_array_equal(TR.get_lat_weights(32, 64, "cpu").numpy(), v["lat_w"])
    # gradient + residual add
    p = torch.from_numpy(v["pred"]).requires_grad_()
    xl = rnd(*p.shape, seed=9)
    T.weighted_mse_loss(xl + p, torch.from_numpy(v["target"]), torch.from_numpy(v["lat_w"]),
                        torch.from_numpy(v["chan_mask"])).backward()
    pd = torch.from_numpy(v["pred"]).to(DEV).requires_grad_()
    TR.weighted_mse_loss(pd, target, lat, chan, x_last=xl.to(DEV)).backward()
    assert rel(pd.grad, p.grad) < TOL


def test_adam_matches_torch(hip):
    from graphcast_lite_amd import hip as H

    p0, g = rnd(1000, seed=1), rnd(1000, seed=2)
    pt = p0.clone().requires_grad_()
    opt = torch.optim.Adam([pt], lr=1e-3)
    p, m, v = p0.to(DEV), torch.zeros(1000, device=DEV), torch.zeros(1000, device=DEV)
    for step in range(1, 6):
        pt.grad = g * step
        opt.step()
        H.adam_step(p, (2 * g * step).to(DEV), m, v, 1e-3, 0.9, 0.999, 1e-8, 0.0, step, grad_scale=0.5)
    assert rel(p, pt) < 1e-6


def _act_ref(x, act, a):
    if act == 0:
        return x
    if act == 1:
        return torch.where(x > 0, x, a * x)
    return x * torch.sigmoid(x)


@pytest.mark.parametrize("rows,Fin,Fout", [(1000, 256, 256), (333, 512, 256), (700, 256, 512), (129, 260, 132),
                                           (5000, 128, 64), (64, 768, 256), (1, 256, 256),
                                           (4500, 260, 132)])  # >= 4096 rows: dX runs on workspace-transposed weights
@pytest.mark.parametrize("act", [0, 1, 2])
def test_dense_fwd_bwd_wide(hip, rows, Fin, Fout, act):
    """gcl_dense_*: wide layers (tiled contraction), SiLU / PReLU on load, weight column blocks (ldw > Fin),
    addend epilogues, gradient column blocks - against fp64 torch autograd."""
    Wfull = rnd(Fout, Fin + 64, seed=2, scale=0.1)
    x = rnd(rows, Fin, seed=1).double().requires_grad_()
    W = Wfull[:, 32:32 + Fin].double().clone().requires_grad_()
    b = rnd(Fout, seed=3).double().requires_grad_()
    add = rnd(rows, Fout, seed=4).double()
    a = torch.tensor([0.25], dtype=torch.float64, requires_grad=True)
    dy = rnd(rows, Fout, seed=5).double()
    y = _act_ref(x, act, a) @ W.t() + b + add
    y.backward(dy)

    Wd = Wfull.to(DEV)[:, 32:32 + Fin]  # a column block: row stride Fin + 64
    xd, bd, addd, dyd = x.detach().float().to(DEV), b.detach().float().to(DEV), add.float().to(DEV), dy.float().to(DEV)
    ad = a.detach().float().to(DEV) if act == 1 else None
    yd = hip.dense_fwd(xd, Wd, bd, act, ad, addend=addd)
    assert rel(yd, y) < TOL
    # dx wrt the pre-activation x, plus an addend (a residual gradient)
    radd = rnd(rows, Fin, seed=6)
    d_slope = torch.zeros(1, device=DEV) if act == 1 else None
    dxd = hip.dense_bwd_dx(dyd, Wd, xd if act else None, act, ad, d_slope, addend=radd.to(DEV))
    assert rel(dxd, x.grad + radd.double()) < TOL
    if act == 1:
        assert abs(d_slope.item() - a.grad.item()) < 1e-4 * max(1.0, abs(a.grad.item()))
    # dW into a column block of a wider gradient, accumulate on top of existing content
    dWfull = torch.ones(Fout, Fin + 64, device=DEV)
    dbd = torch.ones(Fout, device=DEV)
    hip.dense_bwd_dw(dyd, xd, dWfull[:, 32:32 + Fin], dbd, True, act, ad)
    assert rel(dWfull[:, 32:32 + Fin] - 1.0, W.grad) < 5 * TOL
    assert rel(dbd - 1.0, b.grad) < 5 * TOL
    assert (dWfull[:, :32] == 1).all() and (dWfull[:, 32 + Fin:] == 1).all()


def _random_segments(n, E, seed):
    g = torch.Generator().manual_seed(seed)
    rcv = torch.sort(torch.randint(0, n, (E,), generator=g)).values
    rcv[rcv == n // 2] = n // 2 + 1 if n // 2 + 1 < n else 0  # leave at least one empty segment
    rcv = torch.sort(rcv).values
    rowptr = torch.zeros(n + 1, dtype=torch.int64)
    rowptr[1:] = torch.cumsum(torch.bincount(rcv, minlength=n), 0)
    return rcv, rowptr


@pytest.mark.parametrize("B,n,E,D", [(2, 50, 400, 256), (1, 7, 30, 128), (3, 33, 100, 64), (2, 40, 300, 20)])
def test_segment_reduce_and_edge_combine(hip, B, n, E, D):
    """gcl_segment_reduce == scatter(..., reduce="mean"/"sum") on sorted segments (src/models.py:221), with and
    without a permutation; gcl_edge_combine == base + extra + A[ia]*sa[ia] + C[ic]."""
    rcv, rowptr = _random_segments(n, E, 3)
    src = rnd(B, E, D, seed=1)
    ref_mean = P.scatter_mean_rows(src, rcv, n)
    i32 = lambda t: t.to(torch.int32).to(DEV)
    got = hip.segment_reduce(src.to(DEV), None, i32(rowptr), True)
    assert rel(got, ref_mean) < 1e-6
    assert (got.cpu()[:, n // 2] == 0).all()  # empty segment -> exact zeros
    # permuted sum into a column block of a wider output
    perm = torch.randperm(E, generator=torch.Generator().manual_seed(9))
    wide = torch.full((B, n, 2 * D), 5.0, device=DEV)
    hip.segment_reduce(src.to(DEV), i32(perm), i32(rowptr), False, out3=wide[:, :, D:])
    ref_sum = torch.zeros(B, n, D).index_add_(1, rcv, src[:, perm])
    assert rel(wide[:, :, D:], ref_sum) < 1e-6 and (wide[:, :, :D] == 5.0).all()

    A, Cc, base, extra = rnd(B, n, 2 * D, seed=4), rnd(B, n, D, seed=5), rnd(B, E, D, seed=6), rnd(B, E, D, seed=7)
    snd = torch.randint(0, n, (E,), generator=torch.Generator().manual_seed(8))
    sa = torch.rand(n, generator=torch.Generator().manual_seed(10)) + 0.5
    Ad = A.to(DEV)
    got = hip.edge_combine(base.to(DEV), extra.to(DEV), Ad[:, :, D:], i32(snd), sa.to(DEV), Cc.to(DEV), i32(rcv))
    want = base + extra + A[:, snd, D:] * sa[snd].view(1, E, 1) + Cc[:, rcv]
    assert rel(got, want) < 1e-6
    got = hip.edge_combine(None, None, Ad[:, :, :D], i32(snd), None, None, None)
    assert torch.equal(got.cpu(), A[:, snd, :D])


@pytest.mark.parametrize("act", [1, 2])
def test_act_fwd_bwd(hip, act):
    x = rnd(1000, 64, seed=1).double().requires_grad_()
    a = torch.tensor([0.3], dtype=torch.float64, requires_grad=True)
    dy = rnd(1000, 64, seed=2).double()
    y = _act_ref(x, act, a)
    y.backward(dy)
    ad = a.detach().float().to(DEV) if act == 1 else None
    xd = x.detach().float().to(DEV)
    assert rel(hip.act_fwd(xd, act, ad), y) < 1e-6
    ds = torch.zeros(1, device=DEV) if act == 1 else None
    assert rel(hip.act_bwd(xd, dy.float().to(DEV), act, ad, ds), x.grad) < 1e-6
    if act == 1:
        assert abs(ds.item() - a.grad.item()) < 1e-4 * abs(a.grad.item())


@pytest.mark.parametrize("flat", [False, True])
def test_window_pack_matches_loader_bit_for_bit(hip, flat, tmp_path):
    """gcl_window_pack / data.TimeseriesChunkDataset against the numpy restatement of the reference's
    __getitem__ (src/data/dataloader_chunked.py:179-223): bit-identical, both on-disk layouts."""
    import json
    from graphcast_lite_amd.data import TimeseriesChunkDataset
    from oracle.data import window_sample

    rng = np.random.default_rng(7)
    T, n_lon, n_lat, Ct, C, obs, pred = 9, 12, 7, 5, 4, 2, 3
    shape = (T, n_lon * n_lat, Ct) if flat else (T, n_lon, n_lat, Ct)
    series = (rng.standard_normal(shape) * 30).astype(np.float16)
    series.tofile(tmp_path / "data.npy")  # raw memmap, no header
    info = {"n_time": T, "n_feat": Ct, "flat": flat}
    info.update({"n_nodes": n_lon * n_lat} if flat else {"n_lon": n_lon, "n_lat": n_lat})
    (tmp_path / "dataset_info.json").write_text(json.dumps(info))
    mean, std = rng.standard_normal(Ct).astype(np.float64), (rng.random(Ct) + 0.5).astype(np.float64)
    np.savez(tmp_path / "scalers.npz", mean=mean, std=std, n=T)
    ds = TimeseriesChunkDataset(str(tmp_path), obs_window=obs, pred_steps=pred, split="all", n_features=C, device=DEV)
    assert len(ds) == T - obs - pred + 1
    X, Y = ds.batch([3, 0, 4])
    for k, t in enumerate([3, 0, 4]):
        Xr, Yr = window_sample(series, t, obs, pred, C, mean.astype(np.float32)[:C], std.astype(np.float32)[:C], flat)
        assert np.array_equal(X[k].cpu().numpy(), Xr) and np.array_equal(Y[k].cpu().numpy(), Yr)
    x1, y1 = ds[2]
    assert torch.equal(x1, ds.batch([2])[0][0]) and x1.shape == (n_lon * n_lat, obs * C)
    # a window that leaves the series is flagged, not read
    t0 = torch.tensor([T - 2], dtype=torch.int64, device=DEV)
    Xo, Yo = hip.window_pack(ds.chunks[0], t0, ds.mean, ds.std, C, obs, pred)
    assert not torch.isnan(Xo).any() and torch.isnan(Yo).all()
    # batch(out=...) fills caller-owned buffers in place (TrainStep.input_buffers())
    Xa, Ya = ds.batch([3, 0, 4])
    bufs = (torch.full_like(Xa, -7.0), torch.full_like(Ya, -7.0))
    Xb, Yb = ds.batch([3, 0, 4], out=bufs)
    assert Xb.data_ptr() == bufs[0].data_ptr() and torch.equal(Xb, Xa) and torch.equal(Yb, Ya)


@pytest.mark.parametrize("levels,Fin,Fout,B,act", [([1, 2], 64, 64, 3, 1), ([3, 5], 64, 48, 9, 0), ([1, 2], 48, 33, 2, 2),
                                                   ([0], 8, 4, 1, 1)])
def test_gcn_layer_fwd_one_kernel(hip, levels, Fin, Fout, B, act):
    """gcl_gcn_layer_fwd (aggregate-first, one kernel) == the oracle's GCNConv on the activated input, on the
    mesh graph and on the bipartite encoder graph."""
    g = build_graphs(experiment("baseline", mesh_levels=levels))
    a = torch.tensor([0.25])
    for ei, n in ((g["proc"], g["M"]), (g["enc"], g["G"] + g["M"])):
        x, W, b = rnd(B, n, Fin, seed=1), rnd(Fout, Fin, seed=2, scale=0.2), rnd(Fout, seed=3)
        xa = _act_ref(x, act, a)
        ref = torch.stack([P.gcn_conv(xa[i], ei, W, b) for i in range(B)])
        gh = hip.Graph(ei, n, hip.GRAPH_GCN)
        if gh.max_in_degree > 64:  # heavy rows (the 12 icosahedron nodes under 2048 grid points): reported, not run
            with pytest.raises(RuntimeError, match="more than 64 in-edges"):
                hip.gcn_layer_fwd(gh, x.to(DEV), act, a.to(DEV) if act == 1 else None, W.to(DEV), b.to(DEV))
            continue
        got = hip.gcn_layer_fwd(gh, x.to(DEV), act, a.to(DEV) if act == 1 else None, W.to(DEV), b.to(DEV))
        assert rel(got, ref) < TOL


@pytest.mark.parametrize("levels,Fin,Fout,B,act", [([3, 5], 64, 64, 9, 0), ([3, 5], 64, 64, 64, 1), ([3, 5], 64, 64, 3, 2),
                                                   ([2, 4], 48, 33, 5, 1), ([3, 5], 64, 19, 2, 0)])
def test_gcn_layer_fwd_source_tiles(hip, levels, Fin, Fout, B, act, monkeypatch):
    """Mesh rows in tile order: gcl_gcn_layer_fwd stages a tile's source rows once in LDS (gcn_halo_fwd_kernel) instead
    of one gather per edge.  Against the oracle's GCNConv (renaming must be invisible) and BIT-equal to the per-edge
    one-kernel layer (gcn_fwd_kernel), padded 33- / 19-wide outputs included."""
    g = build_graphs(experiment("baseline", mesh_levels=levels))
    n = g["M"]
    ei_t, order, pos = _tiled(g)
    gh = hip.Graph(ei_t, n, hip.GRAPH_GCN)
    assert gh.halo_info(False, 64) is not None
    a = torch.tensor([0.25])
    x, W, b = rnd(B, n, Fin, seed=1), rnd(Fout, Fin, seed=2, scale=0.2), rnd(Fout, seed=3)
    xa = _act_ref(x, act, a)
    ref = torch.stack([P.gcn_conv(xa[i], g["proc"], W, b) for i in range(min(B, 3))])
    outs = {}
    xd = x[:, order].contiguous().to(DEV)
    sl = a.to(DEV) if act == 1 else None
    for mode in ("1", "0"):
        monkeypatch.setenv("GCL_GCN_HALO", mode)
        outs[mode] = hip.gcn_layer_fwd(gh, xd, act, sl, W.to(DEV), b.to(DEV)).cpu()
    if act == hip.ACT_NONE:
        assert torch.equal(outs["1"], outs["0"]), "source-tile layer and per-edge layer differ in some bit"
    else:
        # the staged form activates every staged element ONCE in the LDS image (exactly x > 0 ? x : a x, or the device's
        # SiLU) and then runs the plain sums: bit-equal to the per-edge layer WITHOUT activation on a materialised act(x);
        # the per-edge layer with its own per-edge activation (a w x + (1 - a) w relu(x)) rounds differently
        monkeypatch.setenv("GCL_GCN_HALO", "0")
        pre = hip.gcn_layer_fwd(gh, hip.act_fwd(xd, act, sl), hip.ACT_NONE, None, W.to(DEV), b.to(DEV)).cpu()
        assert torch.equal(outs["1"], pre), "source-tile layer differs in some bit from the per-edge layer on act(x)"
        assert rel(outs["1"], outs["0"]) < 2e-6
    assert rel(outs["1"][: min(B, 3), pos], ref) < TOL


@pytest.mark.parametrize("levels,Fin,Fout,B,act", [([3, 5], 64, 64, 9, 0), ([3, 5], 64, 64, 64, 1), ([2, 4], 48, 33, 3, 2)])
def test_gcn_layer_fwd_through_row_table(hip, levels, Fin, Fout, B, act):
    """gcl_gcn_layer_fwd_tab: the layer's input row i of sample b is x[b, tab[i]] or the batch-invariant flat row
    ~tab[i] of x (the compact pipeline's mesh latents read straight from the encoder output).  BIT-equal to the same
    layer on the materialised rows."""
    g = build_graphs(experiment("baseline", mesh_levels=levels))
    n = g["M"]
    ei_t, order, pos = _tiled(g)
    gh = hip.Graph(ei_t, n, hip.GRAPH_GCN)
    ne = n // 3 + 17
    gen = torch.Generator().manual_seed(5)
    x = rnd(B, ne, Fin, seed=1)
    own = torch.rand(n, generator=gen) < 0.2
    tab = torch.where(own, torch.randint(0, ne, (n,), generator=gen), -torch.randint(0, B * ne, (n,), generator=gen) - 1).to(torch.int32)
    lat = torch.where(own[None, :, None], x[:, tab.clamp(min=0).long()], x.reshape(B * ne, Fin)[(-tab.long() - 1).clamp(min=0)][None])
    a = torch.tensor([0.25])
    W, b = rnd(Fout, Fin, seed=2, scale=0.2), rnd(Fout, seed=3)
    xd = x.to(DEV)
    assert hip.gcn_layer_tab_ok(gh, xd, Fout)
    sl = a.to(DEV) if act == 1 else None
    got = hip.gcn_layer_fwd_tab(gh, xd, tab.to(DEV), act, sl, W.to(DEV), b.to(DEV)).cpu()
    ref = hip.gcn_layer_fwd(gh, lat.contiguous().to(DEV), act, sl, W.to(DEV), b.to(DEV)).cpu()
    assert torch.equal(got, ref), f"{(got != ref).sum().item()} elements differ"


@pytest.mark.parametrize("rows,Fin,Fout", [(1000, 256, 256), (777, 512, 256), (300, 256, 128), (129, 160, 384)])
def test_x3_wide_contractions_exact_on_integers(hip, rows, Fin, Fout):
    """gemm_tile_x3_kernel (csrc/gemm_tile.h: the wide layers of configs[3]/[4] and the InteractionNet MLPs) on small
    integers: every product and partial sum is exact in fp32, so a wrong operand map, piece image or chunk order
    shows up as a wrong integer.  Forward (with bias) and dX must be BIT-equal to the float64 result."""
    g = torch.Generator().manual_seed(11)
    x = torch.randint(-7, 8, (rows, Fin), generator=g).float()
    W = torch.randint(-5, 6, (Fout, Fin), generator=g).float()
    W[::3] += 0.5
    b = torch.randint(-9, 10, (Fout,), generator=g).float()
    dy = torch.randint(-6, 7, (rows, Fout), generator=g).float()
    ref = (x.double() @ W.double().t() + b.double()).float()
    got = hip.dense_fwd(x.to(DEV), W.to(DEV), b.to(DEV), hip.ACT_NONE, None)
    assert torch.equal(got.cpu(), ref), f"forward differs in {(got.cpu() != ref).sum().item()} elements"
    dref = (dy.double() @ W.double()).float()
    dgot = hip.dense_bwd_dx(dy.to(DEV), W.to(DEV), x.to(DEV), hip.ACT_NONE, None, None)
    assert torch.equal(dgot.cpu(), dref), f"dX differs in {(dgot.cpu() != dref).sum().item()} elements"


@pytest.mark.parametrize("n,Fin,Fout,B,wide", [(64 * 9 + 5, 64, 64, 3, False), (300, 48, 33, 2, False), (2000, 64, 19, 5, False),
                                               # neighbours up to 40 rows away: 80 halo rows per 64-row tile, the staged
                                               # kernel's eight-piece halo variant (the icosphere meshes need <= 64)
                                               (64 * 7 + 11, 64, 64, 9, True), (700, 48, 33, 2, True)])
def test_x3_gcn_layer_exact_on_integers(hip, n, Fin, Fout, B, wide, monkeypatch):
    """The one-kernel GCNConv layer with the dense part on the bf16 pipe (gcn_fwd_kernel<.., X3> and, where the graph
    carries a source-tile layout, gcn_halo_fwd_kernel): a ring in which every node has exactly sixteen in-edges (fifteen
    neighbours + its self-loop) gives every edge the weight 1/sqrt(16) * 1/sqrt(16) = 1/16, so with small-integer x and
    W the aggregated tile, every piece product and every partial sum are exact: the layer must be BIT-equal to float64
    (the sixteen edges also walk both halves of the staged form's edge records)."""
    idx = torch.arange(n)
    offs = [-40, -33, -25, -18, -12, -7, -3, -1, 1, 2, 5, 9, 14, 22, 31] if wide else [d for d in range(-7, 9) if d != 0]
    ei = torch.stack([torch.cat([(idx + d) % n for d in offs]), idx.repeat(len(offs))])
    gh = hip.Graph(ei, n, hip.GRAPH_GCN)
    if wide:
        info = gh.halo_info(False, 64)
        assert info is not None and info[2] - 64 > 64, f"expected more than 64 halo rows per tile, got {info}"
    g = torch.Generator().manual_seed(13)
    x = torch.randint(-7, 8, (B, n, Fin), generator=g).float()
    W = torch.randint(-5, 6, (Fout, Fin), generator=g).float()
    W[::3] += 0.5
    b = torch.randint(-9, 10, (Fout,), generator=g).float()
    agg = (x.double() + sum(x.double()[:, (idx + d) % n] for d in offs)) / 16.0  # receiver i: senders i + d and itself
    ref = (agg @ W.double().t() + b.double()).float()
    staged = gh.halo_info(False, 64) is not None and Fin % 16 == 0 and Fin > 32
    for mode in (("1", "0") if staged else ("0",)):
        monkeypatch.setenv("GCL_GCN_HALO", mode)
        got = hip.gcn_layer_fwd(gh, x.to(DEV), hip.ACT_NONE, None, W.to(DEV), b.to(DEV)).cpu()
        assert torch.equal(got, ref), f"GCL_GCN_HALO={mode}: {(got != ref).sum().item()} elements differ"


def test_dense_entry_points_with_zero_rows(hip):
    """An empty set of rows is not an error: outputs are empty, weight gradients are zero (or untouched when
    accumulating)."""
    x, W, b = torch.empty(0, 64, device=DEV), rnd(48, 64, seed=1).to(DEV), rnd(48, seed=2).to(DEV)
    assert tuple(hip.dense_fwd(x, W, b).shape) == (0, 48)
    assert tuple(hip.linear_fwd(x, W, b, None).shape) == (0, 48)
    dy = torch.empty(0, 48, device=DEV)
    assert tuple(hip.dense_bwd_dx(dy, W).shape) == (0, 64)
    dW, db = torch.ones(48, 64, device=DEV), torch.ones(48, device=DEV)
    hip.dense_bwd_dw(dy, x, dW, db, True)
    assert (dW == 1).all() and (db == 1).all()
    hip.dense_bwd_dw(dy, x, dW, db, False)
    assert (dW == 0).all() and (db == 0).all()
