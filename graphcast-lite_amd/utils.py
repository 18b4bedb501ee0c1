"""Static spatial node/edge features (model inputs, part of the layout contract).

Follows `src/utils.py:64-209` (node features: xyz, cos(theta), cos(phi), sin(phi) -> 6 columns)
and `src/utils.py:248-423` + `src/create_graphs.py:37-91` (4-column mesh edge features: length
and receiver-local relative position, both normalised by the longest edge).
"""
import json
from typing import Tuple

import numpy as np
from scipy.spatial.transform import Rotation


def lat_lon_deg_to_spherical(lat: np.ndarray, lon: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    return np.deg2rad(lon), np.deg2rad(90 - lat)


def spherical_to_cartesian(phi: np.ndarray, theta: np.ndarray):
    st = np.sin(theta)
    return np.cos(phi) * st, np.sin(phi) * st, np.cos(theta)


def static_node_features(lat: np.ndarray, lon: np.ndarray) -> np.ndarray:
    """[N,6] = x, y, z, cos(colatitude), cos(lon), sin(lon); dtype follows the inputs."""
    phi, theta = lat_lon_deg_to_spherical(lat, lon)
    x, y, z = spherical_to_cartesian(phi, theta)
    return np.stack([x, y, z, np.cos(theta), np.cos(phi), np.sin(phi)], axis=-1)


def receiver_local_relative_positions(
    phi: np.ndarray, theta: np.ndarray, senders: np.ndarray, receivers: np.ndarray
) -> np.ndarray:
    """Sender position minus receiver position after rotating each receiver to lon=0, lat=0.

    Rotation per receiver: about z by -phi, then about y by (pi/2 - theta)
    (`src/utils.py:326-423`, both local-coordinate switches on)."""
    pos = np.stack(spherical_to_cartesian(phi, theta), axis=-1)
    az = -phi[receivers]
    pol = -theta[receivers] + np.pi / 2
    rot = Rotation.from_euler("zy", np.stack([az, pol], axis=1)).as_matrix()
    r_local = np.einsum("bji,bi->bj", rot, pos[receivers])
    s_local = np.einsum("bji,bi->bj", rot, pos[senders])
    return s_local - r_local


def mesh_edge_features(mesh_lat: np.ndarray, mesh_lon: np.ndarray, edge_index: np.ndarray) -> np.ndarray:
    """[E,4] float32: |rel|/max|rel|, rel/max|rel| (`src/create_graphs.py:37-91`)."""
    phi, theta = lat_lon_deg_to_spherical(mesh_lat, mesh_lon)
    rel = receiver_local_relative_positions(phi, theta, edge_index[0], edge_index[1])
    dist = np.linalg.norm(rel, axis=-1, keepdims=True)
    m = dist.max() if dist.size else 0.0
    if m > 0:
        dist, rel = dist / m, rel / m
    return np.concatenate([dist, rel], axis=-1).astype(np.float32)


def load_from_json_file(path: str):
    with open(path) as fh:
        return json.load(fh)


def save_to_json_file(obj, path: str):
    with open(path, "w") as fh:
        json.dump(obj, fh, indent=2)
