import json
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def lib_built():
    """Make sure libgcl_hip.so exists (hipcc cross-compiles for gfx950 without a GPU)."""
    from graphcast_lite_amd import hip

    if not os.path.exists(hip.LIB_PATH):
        import __graft_entry__ as ge

        ge.build()
    return hip.lib()


# experiment configurations named by BASELINE.json, restated as plain dicts (values taken from the
# reference's experiments/*/config.json; checked against tests/golden/config_parse.json)
def _pipeline(enc_hidden, enc_out, F, proc_type, proc_hidden, dec_mlp_hidden, dec_mlp_out, dec_hidden, out_dim,
              heads=1, enc_dec_type="conv_gcn"):
    gat = {"num_heads": heads, "sparsity_thresholds": [0.0, 0.0]}
    return {
        "encoder": {"mlp": {"mlp_hidden_dims": enc_hidden, "output_dim": enc_out, "use_layer_norm": True,
                            "layer_norm_mode": "node"},
                    "gcn": {"layer_type": enc_dec_type, "hidden_dims": [F, F], "output_dim": F}},
        "processor": {"gcn": {"layer_type": proc_type, "gat_props": gat, "hidden_dims": proc_hidden,
                              "output_dim": F, "use_layer_norm": True, "layer_norm_mode": "node"}},
        "decoder": {"mlp": {"mlp_hidden_dims": dec_mlp_hidden, "output_dim": dec_mlp_out, "use_layer_norm": False},
                    "gcn": {"layer_type": enc_dec_type, "hidden_dims": dec_hidden, "output_dim": out_dim}},
    }


def experiment(name: str, mesh_levels=None):
    from graphcast_lite_amd.config import ExperimentConfig

    graph = {"grid2mesh_edge_creation": "radius", "mesh2grid_edge_creation": "contained",
             "grid2mesh_radius_query": 0.5, "mesh_levels": mesh_levels or [3, 5]}
    data = {"dataset_name": "synthetic", "num_features_used": 33, "obs_window_used": 2, "pred_window_used": 1,
            "want_feats_flattened": True}
    if name == "baseline":
        pipe = _pipeline([48, 48], 64, 64, "conv_gcn", [64, 64], [64, 64], 64, [48, 48], 33)
    elif name == "attention":
        pipe = _pipeline([48, 48], 64, 64, "conv_gat", [64, 64], [64, 64], 64, [48, 48], 33)
    elif name == "attention_h4":
        pipe = _pipeline([48, 48], 64, 64, "conv_gat", [64, 64], [64, 64], 64, [48, 48], 33, heads=4)
    elif name == "sparse_attention":
        pipe = _pipeline([48, 48], 64, 64, "sparse_gat", [], [64, 64], 12, [48, 48], 12, enc_dec_type="simple_conv")
        data.update(num_features_used=12)
    elif name == "wb2_512x256_19f_ar":
        pipe = _pipeline([128, 128], 128, 128, "conv_gcn", [128] * 4, [128, 64], 64, [64, 64], 19)
        graph.update(grid2mesh_radius_query=0.6, mesh_levels=mesh_levels or [4, 6])
        data.update(num_features_used=19)
    else:
        raise KeyError(name)
    return ExperimentConfig(graph=graph, pipeline=pipe, data=data)


def build_graphs(cfg, nlat=32, nlon=64):
    """Graphs + static features through the product's own builder (CPU only)."""
    from graphcast_lite_amd import create_graphs as CG
    from graphcast_lite_amd import mesh as M

    lats = np.linspace(-90, 90, nlat, endpoint=True).astype(np.float32)
    lons = np.linspace(0, 360, nlon, endpoint=False).astype(np.float32)
    meshes = M.get_hierarchy_of_triangular_meshes_for_sphere(max(cfg.graph.mesh_levels))
    fin = meshes[-1]
    mlat, mlon = M.get_mesh_lat_long(fin)
    G = nlat * nlon
    enc, gf, mf = CG.create_encoding_graph(lats, lons, mlat, mlon, fin, cfg.graph, G)
    proc, ef = CG.create_processing_graph(meshes, cfg.graph.mesh_levels, mlat, mlon)
    dec = CG.create_decoding_graph((lats, lons), fin, cfg.graph, G)
    return dict(G=G, M=len(fin.vertices), enc=enc, proc=proc, dec=dec, gfeat=gf, mfeat=mf, efeat=ef, mesh=fin,
                lats=lats, lons=lons)


def oracle_model(cfg, g):
    from oracle import model as omodel

    return omodel.WeatherPrediction(
        cfg.pipeline, cfg.data, num_grid_nodes=g["G"], num_mesh_nodes=g["M"], encoding_graph=g["enc"],
        processing_graph=g["proc"], decoding_graph=g["dec"], init_grid_features=g["gfeat"],
        init_mesh_features=g["mfeat"], processing_edge_features=g["efeat"])


@pytest.fixture(scope="session")
def golden_summary():
    with open(os.path.join(GOLDEN, "graph_summary.json")) as fh:
        return json.load(fh)
