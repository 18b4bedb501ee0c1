"""Graph layout contract: own builder vs fixtures produced by running the reference's graph code
(tests/golden/make_golden.py), the containing-triangle property test (trimesh is absent, so the
reference's create_decoding_graph could not be run), and the host CSR construction."""
import hashlib
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, build_graphs, experiment


def _h(t):
    a = t.numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


@pytest.mark.parametrize("tag,levels", [("L0", [0]), ("L12", [1, 2])])
def test_small_graphs_match_reference_exactly(tag, levels):
    ref = np.load(os.path.join(GOLDEN, f"graph_64x32_{tag}.npz"))
    g = build_graphs(experiment("baseline", mesh_levels=levels))
    assert g["G"] == int(ref["G"]) and g["M"] == int(ref["M"])
    np.testing.assert_array_equal(g["mesh"].vertices, ref["vertices"])
    np.testing.assert_array_equal(g["mesh"].faces, ref["faces"])
    np.testing.assert_array_equal(g["enc"].numpy(), ref["enc_edge_index"])
    np.testing.assert_array_equal(g["proc"].numpy(), ref["proc_edge_index"])
    np.testing.assert_allclose(g["gfeat"].numpy(), ref["grid_static"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(g["mfeat"].numpy(), ref["mesh_static"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(g["efeat"].numpy(), ref["edge_feats"], rtol=0, atol=1e-6)


def test_icosahedron_known_answers():
    # notebooks/src/main.ipynb:178,196: [2,1512] encoder edges, [2,60] mesh edges on the icosahedron
    g = build_graphs(experiment("baseline", mesh_levels=[0]))
    assert tuple(g["enc"].shape) == (2, 1512)
    assert tuple(g["proc"].shape) == (2, 60)
    assert tuple(g["dec"].shape) == (2, 6144)


@pytest.mark.parametrize("tag,name,levels,nlat,nlon", [
    ("64x32_L35", "baseline", [3, 5], 32, 64),
    ("512x256_L46", "wb2_512x256_19f_ar", [4, 6], 256, 512),
])
def test_benchmark_graphs_match_reference_hashes(golden_summary, tag, name, levels, nlat, nlon):
    s = golden_summary[tag]
    g = build_graphs(experiment(name, mesh_levels=levels), nlat, nlon)
    assert (g["G"], g["M"], g["enc"].shape[1], g["proc"].shape[1]) == (s["G"], s["M"], s["E_G2M"], s["E_M"])
    # exact integer layout
    assert _h(g["enc"]) == s["enc_hash"] and _h(g["proc"]) == s["proc_hash"]
    enc = g["enc"].numpy()
    indeg = np.bincount(enc[1] - g["G"], minlength=g["M"])
    hist = {str(k): int(v) for k, v in zip(*np.unique(indeg, return_counts=True))}
    assert hist == s["enc_indeg_hist"]
    # float features: portable checksum
    assert abs(np.abs(g["mfeat"].numpy().astype(np.float64)).sum() - s["mesh_static_checksum"]) < 1e-3
    assert abs(np.abs(g["gfeat"].numpy().astype(np.float64)).sum() - s["grid_static_checksum"]) < 1e-3
    assert abs(np.abs(g["efeat"].numpy().astype(np.float64)).sum() - s["edge_feat_checksum"]) < 1e-2
    # README.md:176: 75 522 mesh edges with self-loops at levels [3,5]
    if tag == "64x32_L35":
        assert g["proc"].shape[1] + g["M"] == 75522


@pytest.mark.parametrize("levels,nlat,nlon", [([0], 32, 64), ([1, 2], 32, 64), ([3, 5], 32, 64), ([4, 6], 64, 128)])
def test_decoding_graph_properties(levels, nlat, nlon):
    """Acceptance for create_decoding_graph (the reference's trimesh call cannot run here):
    receivers are 0,0,0,1,1,1,...; senders are offset by +G; the 3 senders of a grid point are the
    vertices of ONE face of the finest mesh; that face is the one closest to the point in R^3
    (what trimesh.proximity.closest_point returns - src/mesh/grid_mesh_connectivity.py:164-172),
    checked by brute force over all faces; and on fine meshes the point's radial projection lies
    in that face up to the sphere/plane gap."""
    from graphcast_lite_amd.mesh import _closest_point_sqdist, grid_lat_lon_to_xyz

    g = build_graphs(experiment("baseline", mesh_levels=levels), nlat, nlon)
    dec, G = g["dec"].numpy(), g["G"]
    assert dec.shape == (2, 3 * G)
    np.testing.assert_array_equal(dec[1], np.repeat(np.arange(G), 3))
    tri = (dec[0] - G).reshape(G, 3)
    assert tri.min() >= 0 and tri.max() < g["M"]
    F = g["mesh"].faces.astype(np.int64)
    face_id = {tuple(f): i for i, f in enumerate(F.tolist())}
    chosen = np.array([face_id[tuple(t)] for t in tri.tolist()])  # KeyError = not a face
    V = g["mesh"].vertices.astype(np.float64)
    P = grid_lat_lon_to_xyz(g["lats"], g["lons"]).astype(np.float64)
    d_chosen = _closest_point_sqdist(P, V[F[chosen, 0]], V[F[chosen, 1]], V[F[chosen, 2]])
    # brute force: no face is closer than the chosen one
    sample = np.arange(G) if len(F) <= 2000 else np.arange(0, G, max(1, G // 64))
    step = max(1, 2_000_000 // max(len(F), 1))
    for s0 in range(0, len(sample), step):
        ids = sample[s0:s0 + step]
        pp = np.repeat(P[ids], len(F), axis=0)
        ff = np.tile(np.arange(len(F)), len(ids))
        d = _closest_point_sqdist(pp, V[F[ff, 0]], V[F[ff, 1]], V[F[ff, 2]]).reshape(len(ids), len(F))
        assert (d.min(axis=1) >= d_chosen[ids] - 1e-12).all()
    if max(levels) >= 5:
        a, b, c = V[tri[:, 0]], V[tri[:, 1]], V[tri[:, 2]]
        nrm = np.cross(b - a, c - a)
        Q = P * ((nrm * a).sum(1) / (nrm * P).sum(1))[:, None]
        area = (nrm * nrm).sum(1)
        w = np.stack([(np.cross(b - Q, c - Q) * nrm).sum(1), (np.cross(c - Q, a - Q) * nrm).sum(1),
                      (np.cross(a - Q, b - Q) * nrm).sum(1)]) / area
        assert w.min() >= -2e-2


@pytest.mark.parametrize("kind", [0, 1, 2])
def test_host_csr_against_numpy(lib_built, kind):
    from graphcast_lite_amd import hip

    rng = np.random.default_rng(7)
    n, E = 50, 400
    ei = torch.from_numpy(rng.integers(0, n, size=(2, E)).astype(np.int64))
    ei[:, :5] = ei[0, :5]  # a few explicit self-loops
    r = hip.build_csr_host(ei, n, kind)
    s, t = ei[0].numpy(), ei[1].numpy()
    if kind != hip.GRAPH_MEAN:
        keep = s != t
        s, t = np.concatenate([s[keep], np.arange(n)]), np.concatenate([t[keep], np.arange(n)])
    Ep = len(s)
    assert r["num_edges"] == Ep
    order = np.argsort(t, kind="stable")
    np.testing.assert_array_equal(r["col"].numpy(), s[order])
    np.testing.assert_array_equal(r["eperm"].numpy(), order)
    indeg = np.bincount(t, minlength=n)
    np.testing.assert_array_equal(np.diff(r["rowptr"].numpy()), indeg)
    if kind == hip.GRAPH_GCN:
        dis = torch.from_numpy(indeg).float().pow(-0.5).numpy()  # what PyG's gcn_norm computes
        w = (dis[s] * dis[t])[order]
    elif kind == hip.GRAPH_MEAN:
        w = (1.0 / np.maximum(indeg, 1).astype(np.float32))[t][order]
    else:
        w = np.ones(Ep, np.float32)
    np.testing.assert_array_equal(r["w"].numpy(), w.astype(np.float32))
    torder = np.argsort(s, kind="stable")
    np.testing.assert_array_equal(r["tcol"].numpy(), t[torder])
    slot_of = np.empty(Ep, np.int64)
    slot_of[order] = np.arange(Ep)
    np.testing.assert_array_equal(r["tslot"].numpy(), slot_of[torder])
    np.testing.assert_allclose(r["tw"].numpy(), r["w"].numpy()[r["tslot"].numpy()], rtol=0)


def test_host_csr_edge_cases(lib_built):
    from graphcast_lite_amd import hip

    empty = torch.zeros(2, 0, dtype=torch.int64)
    r = hip.build_csr_host(empty, 4, hip.GRAPH_GCN)  # only the appended loops
    assert r["num_edges"] == 4 and r["col"].tolist() == [0, 1, 2, 3] and r["w"].tolist() == [1.0] * 4
    r = hip.build_csr_host(empty, 4, hip.GRAPH_MEAN)
    assert r["num_edges"] == 0 and r["rowptr"].tolist() == [0] * 5
    # single directed edge 0->1: deg=[1,2], y1 = x0/sqrt2 + x1/2, y0 = x0
    r = hip.build_csr_host(torch.tensor([[0], [1]]), 2, hip.GRAPH_GCN)
    np.testing.assert_allclose(r["w"].numpy(), [1.0, 2 ** -0.5, 0.5], rtol=1e-7)


@pytest.mark.parametrize("tag", ["krsk_61x41_L35", "wrap_31x21_L24", "flat_700_L23"])
def test_regional_and_flat_layouts_match_reference(tag):
    """Region-pruned mesh hierarchy (src/mesh/create_mesh.py:225-300, used through
    WeatherPrediction(region_bounds=..., mesh_buffer=...), src/models.py:507-512) and the flat-grid
    encoder graph (create_encoding_graph(flat_grid=True)) against fixtures produced by the reference's code."""
    import json

    from graphcast_lite_amd.config import GraphBuildingConfig
    from graphcast_lite_amd.create_graphs import create_encoding_graph, create_processing_graph
    from graphcast_lite_amd.mesh import (get_hierarchy_of_triangular_meshes_for_sphere, get_mesh_lat_long,
                                         prune_mesh_to_region)

    with open(os.path.join(GOLDEN, "graph_regional.json")) as fh:
        s = json.load(fh)[tag]
    lat_min, lat_max, lon_min, lon_max = s["bounds"]
    if s["flat"]:
        rs = np.random.RandomState(3)
        lats = (lat_min + (lat_max - lat_min) * rs.rand(700)).astype(np.float32)
        lons = (lon_min + (lon_max - lon_min) * rs.rand(700)).astype(np.float32)
    else:
        nlon_r, nlat_r = [int(v) for v in tag.split("_")[1].split("x")]
        lats = np.linspace(lat_min, lat_max, nlat_r).astype(np.float32)
        lons = (np.linspace(lon_min, lon_max, nlon_r) % 360).astype(np.float32)
    assert [float(v) for v in lats[:3]] == s["grid_lats"] and [float(v) for v in lons[:3]] == s["grid_lons"]
    gc = GraphBuildingConfig(grid2mesh_edge_creation="radius", mesh2grid_edge_creation="contained",
                             grid2mesh_radius_query=0.6, mesh_levels=s["levels"])
    meshes = prune_mesh_to_region(get_hierarchy_of_triangular_meshes_for_sphere(splits=max(s["levels"])),
                                  lat_min, lat_max, lon_min, lon_max, buffer_deg=s["buffer"])
    finest = meshes[-1]
    assert len(finest.vertices) == s["M"] and [len(m.faces) for m in meshes] == s["faces_per_level"]
    assert _h(finest.faces) == s["faces_hash"]
    assert abs(np.abs(finest.vertices.astype(np.float64)).sum() - s["vertices_checksum"]) < 1e-4
    mlat, mlon = get_mesh_lat_long(finest)
    enc, gfeat, mfeat = create_encoding_graph(
        grid_node_lats=lats, grid_node_longs=lons, mesh_node_lats=mlat.astype(np.float32),
        mesh_node_longs=mlon.astype(np.float32), mesh=finest, graph_building_config=gc, num_grid_nodes=s["G"],
        flat_grid=s["flat"])
    proc, efeat = create_processing_graph(meshes=meshes, mesh_levels=s["levels"], mesh_node_lats=mlat.astype(np.float32),
                                          mesh_node_longs=mlon.astype(np.float32))
    assert (enc.shape[1], proc.shape[1]) == (s["E_G2M"], s["E_M"])
    assert _h(enc) == s["enc_hash"] and _h(proc) == s["proc_hash"]
    assert abs(np.abs(gfeat.numpy().astype(np.float64)).sum() - s["grid_static_checksum"]) < 1e-3
    assert abs(np.abs(mfeat.numpy().astype(np.float64)).sum() - s["mesh_static_checksum"]) < 1e-3
    assert abs(np.abs(efeat.numpy().astype(np.float64)).sum() - s["edge_feat_checksum"]) < 1e-2


@pytest.mark.parametrize("kind,coef", [("kronecker", (0, 0, 1)), ("cartesian", (1, 1, 0)), ("strong", (1, 1, 1))])
def test_product_graph_equals_dense_kronecker_formula(kind, coef):
    """create_product_graph (built sparsely) == the reference's dense construction
    s01 kron(I_T, A) + s10 kron(A_time, I_N) + s11 kron(A_time, A) -> dense_to_sparse
    (src/models.py:707-774; dense_to_sparse of a 2-D matrix = its non-zero entries in row-major order)."""
    from sklearn.neighbors import kneighbors_graph

    from graphcast_lite_amd.create_graphs import create_product_graph

    lats, lons, T, k = np.linspace(-60, 60, 5), np.linspace(0, 300, 6), 3, 3
    pts = np.array([[a, b] for a in lats for b in lons])
    A = kneighbors_graph(pts, n_neighbors=k, mode="connectivity", include_self=False).toarray()
    N = pts.shape[0]
    Tm = np.zeros((T, T))
    for i in range(T - 1):
        Tm[i, i + 1] = 1
    s01, s10, s11 = coef
    dense = s01 * np.kron(np.eye(T), A) + s10 * np.kron(Tm, np.eye(N)) + s11 * np.kron(Tm, A)
    want = torch.tensor(dense).nonzero().t()
    got = create_product_graph(lats, lons, T, k, kind)
    assert got.dtype == torch.int64 and torch.equal(got, want)
