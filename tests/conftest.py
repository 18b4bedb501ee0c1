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


def experiment(name: str, mesh_levels=None):
    from graphcast_lite_amd.experiments import experiment as _e

    return _e(name, mesh_levels)


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
