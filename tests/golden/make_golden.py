"""Generate golden fixtures by RUNNING the reference's own code in the build container.

Run from the repo root, only where `/root/reference` exists (never on the GPU box):

    python tests/golden/make_golden.py

What runs for real: the reference's icosphere hierarchy, `filter_mesh`,
`get_edges_from_faces`, `radius_query_indices`, `create_encoding_graph`,
`create_processing_graph` (+ edge features), static node features, `weighted_mse_loss`,
`get_lat_weights`, `update_attention_threshold`, and the pydantic parsing of the experiment
configs.  Modules that are absent from this image (`trimesh`, `torch_geometric`, `wandb`) get
inert placeholder entries in `sys.modules` so that the reference's top-level `import` lines
pass; nothing from a placeholder is ever called, so `create_decoding_graph` and every
PyG-backed layer are NOT covered by these fixtures (their parity is "unpinned").

Outputs (data only - arrays and numbers, no reference source text):
  tests/golden/graph_64x32_L0.npz      full edge lists + static feats, mesh levels [0]
  tests/golden/graph_64x32_L12.npz     same, levels [1,2]
  tests/golden/graph_summary.json      shapes, sha256 prefixes and degree histograms, cfg A and B
  tests/golden/graph_regional.json     the same summaries for region-pruned meshes (incl. a box across lon 0/360)
                                       and a flat (per-node coordinates) grid
  tests/golden/loss_vectors.npz        (pred, target, lat_w) -> loss, lat weights, threshold schedule
  tests/golden/config_parse.json       selected fields of the parsed reference configs
  tests/golden/loader_vectors.npz      fp16 series + scalers -> TimeseriesChunkDataset.__getitem__ windows (grid and
                                       flat layouts, every split), src/data/dataloader_chunked.py:33-223
  tests/golden/mlp_vectors.npz         reference MLP (no LayerNorm, src/models.py:54-109): weights, input, output and
                                       autograd gradients for two shapes
  tests/golden/train_loop_vectors.npz  reference train_epoch / test (src/train.py:138-308) + spatial_corr (:114-130)
                                       driven with a stub model (a fixed per-node linear map with `.obs_window`):
                                       AR 1..3, static / forcing channels, masks -> losses, gradients, Adam-updated
                                       weights, (loss, ACC, RMSE) of the evaluation loop
  tests/golden/assemble_vectors.npz    WeatherPrediction._preprocess_input (src/models.py:776-806) on a stub `self`
"""
import hashlib
import json
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def _placeholders():
    class _Inert:  # base class stand-in so `class X(GATConv)` statements evaluate
        def __init__(self, *a, **k):
            raise RuntimeError("placeholder for a module that is absent from this image")

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    mod("trimesh")
    mod("wandb")
    mod("torch_geometric")
    mod("torch_geometric.nn", GCNConv=_Inert, SimpleConv=_Inert, GATConv=_Inert, LayerNorm=_Inert, summary=None)
    mod("torch_geometric.utils", dense_to_sparse=None, softmax=None, scatter=None)


def _h(t) -> str:
    a = t.numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def _graphs(ref, nlat, nlon, levels, radius_q):
    from src.config import GraphBuildingConfig

    gc = GraphBuildingConfig(
        grid2mesh_edge_creation="radius", mesh2grid_edge_creation="contained",
        grid2mesh_radius_query=radius_q, mesh_levels=levels,
    )
    lats = np.linspace(-90, 90, nlat, endpoint=True).astype(np.float32)
    lons = np.linspace(0, 360, nlon, endpoint=False).astype(np.float32)
    meshes = ref["hier"](splits=max(levels))
    finest = meshes[-1]
    mlat, mlon = ref["latlon"](finest_mesh=finest)
    mlat, mlon = mlat.astype(np.float32), mlon.astype(np.float32)
    G = nlat * nlon
    enc, gfeat, mfeat = ref["enc"](
        grid_node_lats=lats, grid_node_longs=lons, mesh_node_lats=mlat, mesh_node_longs=mlon,
        mesh=finest, graph_building_config=gc, num_grid_nodes=G,
    )
    proc, efeat = ref["proc"](meshes=meshes, mesh_levels=levels, mesh_node_lats=mlat, mesh_node_longs=mlon)
    return dict(
        G=G, M=len(finest.vertices), vertices=finest.vertices, faces=finest.faces,
        mesh_lat=mlat, mesh_lon=mlon, enc=enc, proc=proc, gfeat=gfeat, mfeat=mfeat, efeat=efeat,
    )


def main():
    if not os.path.isdir(REF):
        raise SystemExit("reference tree not present; fixtures can only be regenerated in the build container")
    _placeholders()
    sys.path.insert(0, REF)
    from src.create_graphs import create_encoding_graph, create_processing_graph
    from src.mesh.create_mesh import get_hierarchy_of_triangular_meshes_for_sphere
    from src.utils import get_mesh_lat_long

    ref = dict(hier=get_hierarchy_of_triangular_meshes_for_sphere, latlon=get_mesh_lat_long,
               enc=create_encoding_graph, proc=create_processing_graph)

    # --- full small graphs -------------------------------------------------------------
    for tag, levels in (("L0", [0]), ("L12", [1, 2])):
        g = _graphs(ref, 32, 64, levels, 0.5)
        np.savez_compressed(
            os.path.join(HERE, f"graph_64x32_{tag}.npz"),
            G=g["G"], M=g["M"], levels=np.array(levels), radius_q=0.5,
            vertices=g["vertices"], faces=g["faces"], mesh_lat=g["mesh_lat"], mesh_lon=g["mesh_lon"],
            enc_edge_index=g["enc"].numpy(), proc_edge_index=g["proc"].numpy(),
            grid_static=g["gfeat"].numpy(), mesh_static=g["mfeat"].numpy(), edge_feats=g["efeat"].numpy(),
        )

    # --- summaries for the benchmark configs -------------------------------------------
    summary = {}
    for tag, (nlat, nlon, levels, rq) in {
        "64x32_L0": (32, 64, [0], 0.5),
        "64x32_L12": (32, 64, [1, 2], 0.5),
        "64x32_L35": (32, 64, [3, 5], 0.5),
        "512x256_L46": (256, 512, [4, 6], 0.6),
    }.items():
        g = _graphs(ref, nlat, nlon, levels, rq)
        G, M = g["G"], g["M"]
        enc, proc = g["enc"].numpy(), g["proc"].numpy()
        enc_indeg = np.bincount(enc[1] - G, minlength=M)
        proc_indeg = np.bincount(proc[1], minlength=M)
        summary[tag] = dict(
            G=int(G), M=int(M), E_G2M=int(enc.shape[1]), E_M=int(proc.shape[1]),
            enc_hash=_h(g["enc"]), proc_hash=_h(g["proc"]),
            grid_static_hash=_h(g["gfeat"]), mesh_static_hash=_h(g["mfeat"]), edge_feat_hash=_h(g["efeat"]),
            enc_indeg_hist={str(k): int(v) for k, v in zip(*np.unique(enc_indeg, return_counts=True))},
            proc_indeg_hist={str(k): int(v) for k, v in zip(*np.unique(proc_indeg, return_counts=True))},
            enc_sender_sorted=bool((np.diff(enc[0]) >= 0).all()),
            grid_nodes_without_out_edge=int(G - np.unique(enc[0]).shape[0]),
            vertices_checksum=float(np.abs(g["vertices"].astype(np.float64)).sum()),
            mesh_static_checksum=float(np.abs(g["mfeat"].numpy().astype(np.float64)).sum()),
            grid_static_checksum=float(np.abs(g["gfeat"].numpy().astype(np.float64)).sum()),
            edge_feat_checksum=float(np.abs(g["efeat"].numpy().astype(np.float64)).sum()),
            enc_index_checksum=[int(enc[0].astype(np.int64).sum()), int(enc[1].astype(np.int64).sum())],
            proc_index_checksum=[int((proc[0].astype(np.int64) * np.arange(proc.shape[1])).sum()),
                                 int((proc[1].astype(np.int64) * np.arange(proc.shape[1])).sum())],
        )
    with open(os.path.join(HERE, "graph_summary.json"), "w") as fh:
        json.dump(summary, fh, indent=1, sort_keys=True)

    # --- regional (pruned-mesh) and flat-grid layouts (src/main.py:146-173, src/models.py:507-538) ----
    from src.config import GraphBuildingConfig
    from src.mesh.create_mesh import prune_mesh_to_region

    regional = {}
    for tag, (bounds, buf, levels, flat) in {
        "krsk_61x41_L35": ((50.0, 60.0, 85.0, 100.0), 15.0, [3, 5], False),
        "wrap_31x21_L24": ((-10.0, 10.0, 350.0, 365.0), 10.0, [2, 4], False),   # box crossing lon 0/360
        "flat_700_L23": ((40.0, 55.0, 20.0, 45.0), 12.0, [2, 3], True),
    }.items():
        lat_min, lat_max, lon_min, lon_max = bounds
        if flat:  # one (lat, lon) pair per node, irregular spacing
            rs = np.random.RandomState(3)
            lats = (lat_min + (lat_max - lat_min) * rs.rand(700)).astype(np.float32)
            lons = (lon_min + (lon_max - lon_min) * rs.rand(700)).astype(np.float32)
            G = 700
        else:
            nlon_r, nlat_r = [int(v) for v in tag.split("_")[1].split("x")]
            lats = np.linspace(lat_min, lat_max, nlat_r).astype(np.float32)
            lons = (np.linspace(lon_min, lon_max, nlon_r) % 360).astype(np.float32)
            G = nlat_r * nlon_r
        gc = GraphBuildingConfig(grid2mesh_edge_creation="radius", mesh2grid_edge_creation="contained",
                                 grid2mesh_radius_query=0.6, mesh_levels=levels)
        meshes = prune_mesh_to_region(ref["hier"](splits=max(levels)), lat_min, lat_max, lon_min, lon_max, buffer_deg=buf)
        finest = meshes[-1]
        mlat, mlon = ref["latlon"](finest_mesh=finest)
        enc, gfeat, mfeat = ref["enc"](
            grid_node_lats=lats, grid_node_longs=lons, mesh_node_lats=mlat.astype(np.float32),
            mesh_node_longs=mlon.astype(np.float32), mesh=finest, graph_building_config=gc, num_grid_nodes=G,
            flat_grid=flat)
        proc, efeat = ref["proc"](meshes=meshes, mesh_levels=levels, mesh_node_lats=mlat.astype(np.float32),
                                  mesh_node_longs=mlon.astype(np.float32))
        regional[tag] = dict(
            bounds=list(bounds), buffer=buf, levels=levels, flat=flat, G=int(G), M=int(len(finest.vertices)),
            faces_per_level=[int(len(m.faces)) for m in meshes], faces_hash=_h(finest.faces),
            vertices_checksum=float(np.abs(finest.vertices.astype(np.float64)).sum()),
            E_G2M=int(enc.shape[1]), E_M=int(proc.shape[1]), enc_hash=_h(enc), proc_hash=_h(proc),
            grid_static_hash=_h(gfeat), mesh_static_hash=_h(mfeat), edge_feat_hash=_h(efeat),
            grid_static_checksum=float(np.abs(gfeat.numpy().astype(np.float64)).sum()),
            mesh_static_checksum=float(np.abs(mfeat.numpy().astype(np.float64)).sum()),
            edge_feat_checksum=float(np.abs(efeat.numpy().astype(np.float64)).sum()),
            grid_lats=[float(v) for v in lats[:3]], grid_lons=[float(v) for v in lons[:3]],
        )
    with open(os.path.join(HERE, "graph_regional.json"), "w") as fh:
        json.dump(regional, fh, indent=1, sort_keys=True)

    # --- loss / schedule vectors ---------------------------------------------------------
    from src.train import get_lat_weights, update_attention_threshold, weighted_mse_loss, build_boundary_mask

    gen = torch.Generator().manual_seed(1234)
    B, nlat, nlon, C = 3, 32, 64, 5
    pred = torch.randn(B, nlat * nlon, C, generator=gen)
    target = pred + 0.1 * torch.randn(B, nlat * nlon, C, generator=gen)
    lat_w = get_lat_weights(nlat, nlon, "cpu")
    chan = torch.tensor([1.0, 1.0, 0.0, 1.0, 0.5])
    smask = build_boundary_mask(nlon, nlat, 2, "cpu")
    np.savez_compressed(
        os.path.join(HERE, "loss_vectors.npz"),
        pred=pred.numpy(), target=target.numpy(), lat_w=lat_w.numpy(), chan_mask=chan.numpy(),
        spatial_mask=smask.numpy(),
        loss_plain=weighted_mse_loss(pred, target).item(),
        loss_lat=weighted_mse_loss(pred, target, lat_w).item(),
        loss_lat_chan=weighted_mse_loss(pred, target, lat_w, chan).item(),
        loss_all=weighted_mse_loss(pred, target, lat_w, chan, smask).item(),
        thr_epochs=np.arange(0, 40),
        thr_values=np.array([update_attention_threshold(e) for e in range(40)], dtype=np.float64),
    )

    # --- config parsing ------------------------------------------------------------------
    from src.config import ExperimentConfig

    parsed = {}
    for exp in ("baseline", "attention", "sparse_attention", "wb2_512x256_19f_ar"):
        with open(os.path.join(REF, "experiments", exp, "config.json")) as fh:
            cfg = ExperimentConfig(**json.load(fh))
        parsed[exp] = dict(
            mesh_levels=cfg.graph.mesh_levels, radius=cfg.graph.grid2mesh_radius_query,
            enc_mlp_ln=cfg.pipeline.encoder.mlp.use_layer_norm,
            proc_type=cfg.pipeline.processor.gcn.layer_type.value,
            proc_ln=cfg.pipeline.processor.gcn.use_layer_norm,
            dec_mlp_ln=cfg.pipeline.decoder.mlp.use_layer_norm,
            features=cfg.data.num_features_used, obs=cfg.data.obs_window_used,
            flattened=cfg.data.want_feats_flattened, max_ar_steps=cfg.max_ar_steps,
            wandb_log=cfg.wandb_log,
        )
    with open(os.path.join(HERE, "config_parse.json"), "w") as fh:
        json.dump(parsed, fh, indent=1, sort_keys=True)
    _loader_vectors()
    _mlp_vectors()
    _train_loop_vectors()
    _assemble_vectors()
    print("golden fixtures written to", HERE)


class StubModel(torch.nn.Module):
    """What src/train.py needs from a model: `.obs_window` and `model(X=, attention_threshold=, epoch=,
    batch_num=) -> [N, G, C]`.  A fixed per-node linear map of the flattened window (+ tanh so that
    the AR gradient is not trivially linear).  Test scaffolding of THIS repo, not reference code."""

    def __init__(self, obs, C, seed):
        super().__init__()
        self.obs_window = obs
        g = torch.Generator().manual_seed(seed)
        self.W = torch.nn.Parameter(0.3 * torch.randn(C, obs * C, generator=g))
        self.b = torch.nn.Parameter(0.1 * torch.randn(C, generator=g))

    def forward(self, X, attention_threshold=0.0, **kw):
        return 0.5 * torch.tanh(X @ self.W.t() + self.b)


def _loader_vectors():
    """Run the reference's TimeseriesChunkDataset on small synthetic fp16 series written to a temp dir."""
    import tempfile

    from src.data.dataloader_chunked import TimeseriesChunkDataset

    out = {}
    rs = np.random.RandomState(11)
    with tempfile.TemporaryDirectory() as d:
        # regular grid (T, lon, lat, Ct) as a headerless memmap + dataset_info.json
        T, n_lon, n_lat, Ct, C = 14, 6, 5, 7, 5
        series = (rs.randn(T, n_lon, n_lat, Ct) * 3 + 1).astype(np.float16)
        mean, std = rs.randn(Ct).astype(np.float64), (0.5 + rs.rand(Ct)).astype(np.float64)
        gd = os.path.join(d, "grid")
        os.makedirs(gd)
        series.tofile(os.path.join(gd, "data.npy"))
        np.savez(os.path.join(gd, "scalers.npz"), mean=mean, std=std, n=T)
        with open(os.path.join(gd, "dataset_info.json"), "w") as fh:
            json.dump({"n_time": T, "n_lon": n_lon, "n_lat": n_lat, "n_feat": Ct}, fh)
        out.update(grid_series=series, grid_mean=mean, grid_std=std, grid_C=C)
        for split in ("train", "test", "val", "test_only", "all"):
            ds = TimeseriesChunkDataset(gd, obs_window=2, pred_steps=3, split=split, n_features=C)
            out[f"grid_{split}_len"] = len(ds)
            out[f"grid_{split}_t0"] = np.array([t for _, t in ds._sample_indices], dtype=np.int64)
            if len(ds):
                X, Y = zip(*[ds[i] for i in range(len(ds))])
                out[f"grid_{split}_X"] = torch.stack(X).numpy()
                out[f"grid_{split}_Y"] = torch.stack(Y).numpy()
        # flat nodes (T, N, Ct)
        T2, N, Ct2 = 9, 23, 4
        fseries = (rs.randn(T2, N, Ct2) * 2).astype(np.float16)
        fmean, fstd = rs.randn(Ct2).astype(np.float64), (0.5 + rs.rand(Ct2)).astype(np.float64)
        fd = os.path.join(d, "flat")
        os.makedirs(fd)
        fseries.tofile(os.path.join(fd, "data.npy"))
        np.savez(os.path.join(fd, "scalers.npz"), mean=fmean, std=fstd, n=T2)
        with open(os.path.join(fd, "dataset_info.json"), "w") as fh:
            json.dump({"n_time": T2, "n_nodes": N, "n_feat": Ct2, "flat": True}, fh)
        ds = TimeseriesChunkDataset(fd, obs_window=3, pred_steps=1, split="all")
        X, Y = zip(*[ds[i] for i in range(len(ds))])
        out.update(flat_series=fseries, flat_mean=fmean, flat_std=fstd, flat_all_X=torch.stack(X).numpy(),
                   flat_all_Y=torch.stack(Y).numpy(),
                   flat_all_t0=np.array([t for _, t in ds._sample_indices], dtype=np.int64))
        # legacy multi-chunk layout (chunk_*.npy with headers): windows never cross a chunk boundary
        cd = os.path.join(d, "chunks")
        os.makedirs(cd)
        c0, c1 = series[:6], series[6:]
        np.save(os.path.join(cd, "chunk_0.npy"), c0)
        np.save(os.path.join(cd, "chunk_1.npy"), c1)
        np.savez(os.path.join(cd, "scalers.npz"), mean=mean, std=std, n=T)
        ds = TimeseriesChunkDataset(cd, obs_window=2, pred_steps=1, split="all", n_features=C)
        out["chunks_all_index"] = np.array(ds._sample_indices, dtype=np.int64)
        X, Y = zip(*[ds[i] for i in range(len(ds))])
        out["chunks_all_X"], out["chunks_all_Y"] = torch.stack(X).numpy(), torch.stack(Y).numpy()
    np.savez_compressed(os.path.join(HERE, "loader_vectors.npz"), **out)


def _mlp_vectors():
    """The reference's pure-torch MLP (no LayerNorm: PyG's is absent) with saved weights."""
    from src.config import MLPBlock
    from src.models import MLP

    out = {}
    for tag, (rows, fin, hidden, fout) in {"enc": (300, 72, [48, 48], 64), "dec": (257, 64, [64, 64], 64),
                                           "odd": (50, 19, [33], 19), "single": (64, 20, [], 12)}.items():
        torch.manual_seed(5)
        m = MLP(MLPBlock(mlp_hidden_dims=hidden, output_dim=fout, use_layer_norm=False), input_dim=fin)
        g = torch.Generator().manual_seed(17)
        with torch.no_grad():
            for p in m.parameters():
                if p.numel() == 1:
                    p.fill_(0.1 + 0.3 * float(torch.rand(1, generator=g)))  # distinct PReLU slopes
        x = torch.randn(rows, fin, generator=g, requires_grad=True)
        dy = torch.randn(rows, fout, generator=g)
        y = m(x)
        y.backward(dy)
        out[f"{tag}_x"], out[f"{tag}_dy"], out[f"{tag}_y"], out[f"{tag}_dx"] = (x.detach().numpy(), dy.numpy(),
                                                                              y.detach().numpy(), x.grad.numpy())
        out[f"{tag}_keys"] = np.array(list(m.state_dict().keys()))
        for k, v in m.state_dict().items():
            out[f"{tag}_w_{k}"] = v.numpy()
        for k, p in m.named_parameters():
            out[f"{tag}_g_{k}"] = p.grad.numpy()
    np.savez_compressed(os.path.join(HERE, "mlp_vectors.npz"), **out)


def _train_loop_vectors():
    from src.train import (build_boundary_mask, get_lat_weights, spatial_corr, test, train_epoch)

    out = {}
    n_lat, n_lon, obs, C = 6, 8, 2, 5
    G = n_lat * n_lon
    gen = torch.Generator().manual_seed(101)
    batches = [(torch.randn(2, G, obs * C, generator=gen), 0.7 * torch.randn(2, G, 3 * C, generator=gen)) for _ in range(3)]
    out["X"] = torch.stack([b[0] for b in batches]).numpy()
    out["Y"] = torch.stack([b[1] for b in batches]).numpy()
    lat_w = get_lat_weights(n_lat, n_lon, "cpu")
    chan = torch.tensor([1.0, 1.0, 0.0, 1.0, 0.5])
    smask = build_boundary_mask(n_lon, n_lat, 1, "cpu")
    out.update(lat_w=lat_w.numpy(), chan_mask=chan.numpy(), spatial_mask=smask.numpy(), obs=obs, C=C, n_lat=n_lat, n_lon=n_lon)
    cases = {
        "ar1_plain": dict(current_ar_steps=1),
        "ar1_lat": dict(current_ar_steps=1, lat_weights=lat_w),
        "ar2_static_forcing": dict(current_ar_steps=2, lat_weights=lat_w, channel_mask=chan, static_channels=[2],
                                   forcing_channels=[4]),
        "ar3_all": dict(current_ar_steps=3, lat_weights=lat_w, channel_mask=chan, spatial_mask=smask,
                        static_channels=[2], forcing_channels=[0, 4]),
        "ar3_noresidual": dict(current_ar_steps=3, lat_weights=lat_w, static_channels=[2], forcing_channels=[4],
                               use_residual=False),
        "ar5_capped": dict(current_ar_steps=5, lat_weights=lat_w),  # more steps asked than targets exist: runs 3
    }
    out["cases"] = np.array(list(cases))
    for tag, kw in cases.items():
        m = StubModel(obs, C, seed=9)
        if tag == "ar1_plain":
            out["W0"], out["b0"] = m.W.detach().numpy().copy(), m.b.detach().numpy().copy()
        opt = torch.optim.Adam(m.parameters(), lr=1e-2)
        losses = [train_epoch(m, batches, opt, None, "cpu", 0.0, ep, **kw) for ep in range(2)]
        out[f"{tag}_epoch_losses"] = np.array(losses, dtype=np.float64)
        out[f"{tag}_W"], out[f"{tag}_b"] = m.W.detach().numpy(), m.b.detach().numpy()
        out[f"{tag}_gW_last"], out[f"{tag}_gb_last"] = m.W.grad.numpy().copy(), m.b.grad.numpy().copy()
        # gradient of the FIRST batch at the initial weights (no optimiser in between)
        m0 = StubModel(obs, C, seed=9)
        o0 = torch.optim.SGD(m0.parameters(), lr=0.0)
        out[f"{tag}_loss_first"] = train_epoch(m0, batches[:1], o0, None, "cpu", 0.0, 0, **kw)
        out[f"{tag}_gW_first"], out[f"{tag}_gb_first"] = m0.W.grad.numpy().copy(), m0.b.grad.numpy().copy()
    # evaluation loop (one-step, carry-forward, weighted MSE / spatial ACC / raw RMSE)
    m = StubModel(obs, C, seed=9)
    ev = {
        "eval_plain": dict(),
        "eval_all": dict(lat_weights=lat_w, spatial_mask=smask, channel_mask=chan, static_channels=[2], forcing_channels=[0, 4]),
        "eval_noresidual": dict(lat_weights=lat_w, static_channels=[2], use_residual=False),
    }
    for tag, kw in ev.items():
        out[tag] = np.array(test(m, batches, None, "cpu", **kw), dtype=np.float64)
    # single-target batches (y is [N, G, C]: the `total_target_steps == 1` branch)
    b1 = [(X, Y[..., :C].contiguous()) for X, Y in batches]
    out["eval_single_target"] = np.array(test(m, b1, None, "cpu", lat_weights=lat_w), dtype=np.float64)
    # spatial_corr on its own
    p, t = torch.randn(3, G, C, generator=gen), torch.randn(3, G, C, generator=gen)
    out["sc_pred"], out["sc_true"] = p.numpy(), t.numpy()
    out["sc_batched"] = spatial_corr(p, t)
    out["sc_sample0"] = spatial_corr(p[0], t[0])
    out["sc_excl"] = spatial_corr(p, t, exclude_channels=[1, 3])
    out["sc_const"] = spatial_corr(torch.ones(G, C), t[0])  # zero variance: the +1e-8 guard
    np.savez_compressed(os.path.join(HERE, "train_loop_vectors.npz"), **out)


def _assemble_vectors():
    from src.models import WeatherPrediction

    out = {}
    gen = torch.Generator().manual_seed(3)
    for tag, (G, M, Cdyn, Cs, product) in {"a": (37, 12, 10, 6, False), "b": (16, 40, 7, 6, True)}.items():
        stub = types.SimpleNamespace(
            init_grid_features=torch.randn(G, Cs, generator=gen), init_mesh_features=torch.randn(M, Cs, generator=gen),
            num_features=Cdyn, total_feature_size=Cdyn if not product else 3 * Cdyn, use_product_graph=product,
            _num_mesh_nodes=M, device="cpu")
        x = torch.randn(G, Cdyn, generator=gen)
        y = WeatherPrediction._preprocess_input(stub, x)
        out[f"{tag}_x"], out[f"{tag}_gs"], out[f"{tag}_ms"], out[f"{tag}_out"] = (
            x.numpy(), stub.init_grid_features.numpy(), stub.init_mesh_features.numpy(), y.numpy())
    np.savez_compressed(os.path.join(HERE, "assemble_vectors.npz"), **out)


if __name__ == "__main__":
    main()
