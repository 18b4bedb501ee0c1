"""Encoding / processing / decoding graphs in the reference's layout.

Same signatures and return values as `src/create_graphs.py:96-295`: `edge_index` is int64
`[2,E]`, row 0 = sender, row 1 = receiver; nodes are numbered grid first (0..G-1) then mesh
(G..G+M-1); grid points flatten lat-major.
"""
from typing import List, Tuple

import numpy as np
import torch

from .config import GraphBuildingConfig, Grid2MeshEdgeCreation, Mesh2GridEdgeCreation
from .mesh import (
    TriangularMesh,
    filter_mesh,
    get_edges_from_faces,
    get_max_edge_distance,
    in_mesh_triangle_indices,
    radius_query_indices,
)
from .utils import mesh_edge_features, static_node_features


def create_encoding_graph(
    grid_node_lats: np.ndarray,
    grid_node_longs: np.ndarray,
    mesh_node_lats: np.ndarray,
    mesh_node_longs: np.ndarray,
    mesh: TriangularMesh,
    graph_building_config: GraphBuildingConfig,
    num_grid_nodes: int,
    flat_grid: bool = False,
) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """E_G2M (`src/create_graphs.py:96-196`): every mesh vertex within
    `radius_query * longest finest-mesh edge` of a grid point receives from it.
    Returns (edge_index, grid static feats [G,6], mesh static feats [M,6])."""
    if graph_building_config.grid2mesh_edge_creation != Grid2MeshEdgeCreation.RADIUS:
        raise NotImplementedError(
            f"There is no support for {graph_building_config.grid2mesh_edge_creation} to create Grid2Mesh edges."
        )
    radius = get_max_edge_distance(mesh) * graph_building_config.grid2mesh_radius_query
    g_idx, m_idx = radius_query_indices(
        grid_latitude=grid_node_lats, grid_longitude=grid_node_longs, mesh=mesh, radius=radius, flat=flat_grid
    )
    edge_index = torch.from_numpy(np.stack([g_idx, m_idx + num_grid_nodes]).astype(np.int64))

    if flat_grid:
        glat = np.asarray(grid_node_lats).reshape(-1).astype(np.float32)
        glon = np.asarray(grid_node_longs).reshape(-1).astype(np.float32)
    else:
        glon2, glat2 = np.meshgrid(grid_node_longs, grid_node_lats)
        glon, glat = glon2.reshape(-1).astype(np.float32), glat2.reshape(-1).astype(np.float32)
    grid_feats = static_node_features(glat, glon)
    mesh_feats = static_node_features(mesh_node_lats, mesh_node_longs)
    return (
        edge_index,
        torch.tensor(grid_feats, dtype=torch.float32),
        torch.tensor(mesh_feats, dtype=torch.float32),
    )


def create_processing_graph(
    meshes: List[TriangularMesh],
    mesh_levels: List[int],
    mesh_node_lats: np.ndarray = None,
    mesh_node_longs: np.ndarray = None,
):
    """E_M (`src/create_graphs.py:199-240`): undirected edges of all selected levels' faces.
    With coordinates also returns the [E,4] edge features."""
    merged = filter_mesh(meshes=meshes, mesh_levels=mesh_levels)
    edge_index = torch.tensor(get_edges_from_faces(merged.faces), dtype=torch.int64)
    if mesh_node_lats is not None and mesh_node_longs is not None:
        feats = mesh_edge_features(mesh_node_lats, mesh_node_longs, edge_index.numpy())
        return edge_index, torch.from_numpy(feats)
    return edge_index


def create_decoding_graph(
    cordinates: Tuple[np.ndarray, np.ndarray],
    mesh: TriangularMesh,
    graph_building_config: GraphBuildingConfig,
    num_grid_nodes: int,
    flat_grid: bool = False,
) -> torch.Tensor:
    """E_M2G (`src/create_graphs.py:244-295`): each grid point receives from the 3 vertices of
    its containing finest-mesh face; receivers are 0,0,0,1,1,1,..."""
    if graph_building_config.mesh2grid_edge_creation != Mesh2GridEdgeCreation.CONTAINED:
        raise NotImplementedError(
            f"There is no support for {graph_building_config.mesh2grid_edge_creation} to create Mesh2Grid edges."
        )
    g_idx, m_idx = in_mesh_triangle_indices(
        grid_latitude=cordinates[0], grid_longitude=cordinates[1], mesh=mesh, flat=flat_grid
    )
    return torch.from_numpy(np.stack([m_idx + num_grid_nodes, g_idx]).astype(np.int64))


def create_product_graph(grid_lat: np.ndarray, grid_lon: np.ndarray, obs_window: int, num_k: int, graph_type: str):
    """Time x space product graph of `WeatherPrediction._create_product_graph` (`src/models.py:707-774`):
    a directed chain over the T observed steps, a k-nearest-neighbour graph over the (lat, lon) grid
    points (`sklearn.neighbors.kneighbors_graph`, connectivity, no self), combined as
        s01 (I_T (x) A_space) + s10 (A_time (x) I_N) + s11 (A_time (x) A_space)
    with (s01, s10, s11) = (0,0,1) kronecker / (1,1,0) cartesian / (1,1,1) strong (s00 is always 0, so
    `self_loop` has no effect there either).  Returned as `dense_to_sparse` would: the non-zero entries
    in row-major order, row index first.  Built sparsely - the dense (T N)^2 matrix is never formed."""
    from sklearn.neighbors import kneighbors_graph

    kind = getattr(graph_type, "value", graph_type)
    s01, s10, s11 = {"kronecker": (0, 0, 1), "cartesian": (1, 1, 0), "strong": (1, 1, 1)}[kind]
    pts = np.array([[lat, lon] for lat in grid_lat for lon in grid_lon])
    A = kneighbors_graph(pts, n_neighbors=num_k, mode="connectivity", include_self=False).tocoo()
    ar, ac = A.row.astype(np.int64), A.col.astype(np.int64)
    T, N = int(obs_window), pts.shape[0]
    rows, cols = [], []
    t_all, n_all = np.arange(T, dtype=np.int64), np.arange(N, dtype=np.int64)
    if s01:  # same step, spatial neighbours
        rows.append((t_all[:, None] * N + ar[None, :]).reshape(-1))
        cols.append((t_all[:, None] * N + ac[None, :]).reshape(-1))
    t_from = np.arange(T - 1, dtype=np.int64)  # temporal chain i -> i + 1
    if s10:  # same node, next step
        rows.append((t_from[:, None] * N + n_all[None, :]).reshape(-1))
        cols.append(((t_from[:, None] + 1) * N + n_all[None, :]).reshape(-1))
    if s11:  # next step, spatial neighbours
        rows.append((t_from[:, None] * N + ar[None, :]).reshape(-1))
        cols.append(((t_from[:, None] + 1) * N + ac[None, :]).reshape(-1))
    r, c = np.concatenate(rows), np.concatenate(cols)
    key = np.unique(r * (T * N) + c)  # union of the terms (all coefficients are positive), row-major order
    return torch.from_numpy(np.stack([key // (T * N), key % (T * N)]).astype(np.int64))
