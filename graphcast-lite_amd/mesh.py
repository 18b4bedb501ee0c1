"""Icosphere multi-mesh and grid<->mesh geometric queries (one-time CPU setup).

This is the layout contract of SURVEY.md §8a row 11: it must emit the same vertices, faces and
index orderings as the reference so that graphs (and state-dicts trained on them) line up.

Reference behaviour followed (not copied; everything here is array-at-a-time numpy):
  * icosahedron + 4-way face split, child vertex numbering by first appearance while walking
    faces in order and edges (v0v1, v1v2, v2v0) - `src/mesh/create_mesh.py:75-207`
  * level merge (finest first)                               - `src/mesh/create_mesh.py:210-223`
  * undirected edge list from faces                          - `src/mesh/create_mesh.py:323-352`
  * regional pruning                                         - `src/mesh/create_mesh.py:225-320`
  * radius query (cKDTree ball query in R^3 chord distance)  - `src/mesh/grid_mesh_connectivity.py:53-104`
  * containing-triangle query                                - `src/mesh/grid_mesh_connectivity.py:139-184`
    (the reference delegates to `trimesh.proximity.closest_point`; trimesh is absent here, so
    the closest-face search is implemented directly: exact point-triangle distance over the
    faces incident to the nearest vertices, ties to the lowest face index)
  * lat/lon of mesh vertices                                 - `src/utils.py:212-245,426-437`
"""
from typing import List, NamedTuple, Sequence, Tuple

import numpy as np
from scipy.spatial import cKDTree
from scipy.spatial.transform import Rotation


class TriangularMesh(NamedTuple):
    """vertices [V,3] float32 on the unit sphere; faces [F,3] int32, counter-clockwise from outside."""

    vertices: np.ndarray
    faces: np.ndarray


# 20 faces of the icosahedron in the reference's order (`src/mesh/create_mesh.py:141-162`); the
# order is part of the layout contract because child-vertex numbering depends on it.
_ICO_FACES = np.array(
    [
        (0, 1, 2), (0, 6, 1), (8, 0, 2), (8, 4, 0), (3, 8, 2), (3, 2, 7), (7, 2, 1),
        (0, 4, 6), (4, 11, 6), (6, 11, 5), (1, 5, 7), (4, 10, 11), (4, 8, 10), (10, 8, 3),
        (10, 3, 9), (11, 10, 9), (11, 9, 5), (5, 9, 7), (9, 3, 7), (1, 6, 5),
    ],
    dtype=np.int32,
)


def get_icosahedron() -> TriangularMesh:
    """Regular icosahedron with unit circumsphere, rotated so that two faces are polar."""
    g = (1.0 + np.sqrt(5.0)) / 2.0
    rows = []
    for s1 in (1.0, -1.0):
        for s2 in (g, -g):
            rows += [(s1, s2, 0.0), (0.0, s1, s2), (s2, 0.0, s1)]
    v = np.asarray(rows, dtype=np.float32)
    v /= np.linalg.norm([1.0, g])
    dihedral = 2.0 * np.arcsin(g / np.sqrt(3.0))
    rot = Rotation.from_euler(seq="y", angles=(np.pi - dihedral) / 2.0).as_matrix()
    v = np.dot(v, rot)
    return TriangularMesh(vertices=v.astype(np.float32), faces=_ICO_FACES.copy())


def _split_faces(mesh: TriangularMesh) -> TriangularMesh:
    """One 4-way subdivision; new vertices are edge midpoints pushed back onto the sphere."""
    f = mesh.faces.astype(np.int64)
    nv = mesh.vertices.shape[0]
    # the 3 edges of every face in walk order: (0,1) (1,2) (2,0) per face, faces in order
    a = f[:, [0, 1, 2]].reshape(-1)
    b = f[:, [1, 2, 0]].reshape(-1)
    key = np.minimum(a, b) * nv + np.maximum(a, b)
    uniq, first, inv = np.unique(key, return_index=True, return_inverse=True)
    # number children by first appearance in the walk
    order = np.argsort(first, kind="stable")
    rank = np.empty_like(order)
    rank[order] = np.arange(order.shape[0])
    child = (nv + rank[inv]).reshape(-1, 3)  # [F,3]: mid(01), mid(12), mid(20)

    pa = a[first[order]]
    pb = b[first[order]]
    # float32 midpoint then float32 normalisation, as the per-vertex reference arithmetic does
    mid = (mesh.vertices[pa] + mesh.vertices[pb]) / np.float32(2.0)
    # per-row BLAS dot: the reference normalises vertex by vertex with `np.linalg.norm`, whose
    # 1-D path is sqrt(x.dot(x)); a vectorised sum-of-squares differs from it in the last ulp
    nrm = np.sqrt(np.fromiter((r.dot(r) for r in mid), dtype=np.float32, count=mid.shape[0]))
    mid = (mid / nrm[:, None]).astype(np.float32)
    verts = np.concatenate([mesh.vertices, mid], axis=0)

    v0, v1, v2 = f[:, 0], f[:, 1], f[:, 2]
    m01, m12, m20 = child[:, 0], child[:, 1], child[:, 2]
    new_faces = np.stack(
        [
            np.stack([v0, m01, m20], 1),
            np.stack([m01, v1, m12], 1),
            np.stack([m20, m12, v2], 1),
            np.stack([m01, m12, m20], 1),
        ],
        axis=1,
    ).reshape(-1, 3)
    return TriangularMesh(vertices=verts, faces=new_faces.astype(np.int32))


def get_hierarchy_of_triangular_meshes_for_sphere(splits: int) -> List[TriangularMesh]:
    """Meshes for levels 0..splits; level k+1's first V_k vertices are level k's vertices."""
    out = [get_icosahedron()]
    for _ in range(splits):
        out.append(_split_faces(out[-1]))
    return out


def filter_mesh(meshes: Sequence[TriangularMesh], mesh_levels: Sequence[int]) -> TriangularMesh:
    """Multi-mesh: vertices of the finest requested level, faces of all requested levels
    concatenated finest-first."""
    levels = sorted(mesh_levels, reverse=True)
    faces = np.concatenate([meshes[lv].faces for lv in levels], axis=0)
    return TriangularMesh(vertices=meshes[levels[0]].vertices, faces=faces)


def get_edges_from_faces(faces: np.ndarray) -> np.ndarray:
    """[2, 2*U] int: unique undirected edges sorted by (min,max), each followed by its reverse."""
    f = np.asarray(faces).astype(np.int64)
    a = f[:, [0, 1, 2]].reshape(-1)
    b = f[:, [1, 2, 0]].reshape(-1)
    lo, hi = np.minimum(a, b), np.maximum(a, b)
    big = int(hi.max()) + 1 if hi.size else 1
    key = np.unique(lo * big + hi)
    lo, hi = key // big, key % big
    out = np.empty((2, 2 * key.shape[0]), dtype=np.asarray(faces).dtype)
    out[0, 0::2], out[1, 0::2] = lo, hi
    out[0, 1::2], out[1, 1::2] = hi, lo
    return out


def faces_to_edges(faces: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    f = np.asarray(faces)
    return f.T.reshape(-1), f[:, [1, 2, 0]].T.reshape(-1)


def get_max_edge_distance(mesh: TriangularMesh) -> float:
    s, r = faces_to_edges(mesh.faces)
    return np.linalg.norm(mesh.vertices[s] - mesh.vertices[r], axis=-1).max()


def cartesian_to_lat_lon(xyz: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """Latitude [-90,90] and longitude [0,360) in degrees of unit vectors (input dtype kept)."""
    phi = np.arctan2(xyz[:, 1], xyz[:, 0])
    with np.errstate(invalid="ignore"):
        theta = np.arccos(xyz[:, 2])
    lon = np.mod(np.rad2deg(phi), 360)
    lat = 90 - np.rad2deg(theta)
    return lat, lon


def get_mesh_lat_long(finest_mesh: TriangularMesh) -> Tuple[np.ndarray, np.ndarray]:
    lat, lon = cartesian_to_lat_lon(finest_mesh.vertices)
    return lat.astype(np.float32), lon.astype(np.float32)


def grid_lat_lon_to_xyz(lat: np.ndarray, lon: np.ndarray, flat: bool = False) -> np.ndarray:
    """[G,3] unit vectors; regular grids flatten lat-major (lat slow, lon fast)."""
    if not flat:
        lon, lat = np.meshgrid(lon, lat)
    phi = np.deg2rad(lon)
    theta = np.deg2rad(90 - lat)
    st = np.sin(theta)
    return np.stack([np.cos(phi) * st, np.sin(phi) * st, np.cos(theta)], axis=-1).reshape(-1, 3)


def radius_query_indices(
    *, grid_latitude, grid_longitude, mesh: TriangularMesh, radius: float, flat: bool = False
) -> Tuple[np.ndarray, np.ndarray]:
    """All (grid point, mesh vertex) pairs within `radius` (chord distance), grouped by grid
    point in ascending order with ascending mesh index inside a group."""
    pts = grid_lat_lon_to_xyz(grid_latitude, grid_longitude, flat)
    hits = cKDTree(mesh.vertices).query_ball_point(x=pts, r=radius, return_sorted=True)
    counts = np.fromiter((len(h) for h in hits), dtype=np.int64, count=len(hits))
    grid_idx = np.repeat(np.arange(len(hits)), counts).astype(int)
    mesh_idx = (
        np.concatenate([np.asarray(h, dtype=int) for h in hits]) if counts.sum() else np.zeros(0, int)
    )
    return grid_idx, mesh_idx


def _closest_point_sqdist(p: np.ndarray, a: np.ndarray, b: np.ndarray, c: np.ndarray) -> np.ndarray:
    """Squared distance from points p [N,3] to triangles (a,b,c) [N,3] each (Ericson's regions)."""
    ab, ac, ap = b - a, c - a, p - a
    d1 = (ab * ap).sum(-1)
    d2 = (ac * ap).sum(-1)
    bp = p - b
    d3 = (ab * bp).sum(-1)
    d4 = (ac * bp).sum(-1)
    cp = p - c
    d5 = (ab * cp).sum(-1)
    d6 = (ac * cp).sum(-1)
    vc = d1 * d4 - d3 * d2
    vb = d5 * d2 - d1 * d6
    va = d3 * d6 - d5 * d4
    with np.errstate(divide="ignore", invalid="ignore"):
        den = va + vb + vc
        v_in = vb / den
        w_in = vc / den
        t_ab = d1 / (d1 - d3)
        t_ac = d2 / (d2 - d6)
        t_bc = (d4 - d3) / ((d4 - d3) + (d5 - d6))
    q = a + ab * v_in[:, None] + ac * w_in[:, None]  # interior
    m = (va <= 0) & ((d4 - d3) >= 0) & ((d5 - d6) >= 0)
    q = np.where(m[:, None], b + (c - b) * t_bc[:, None], q)
    m = (vb <= 0) & (d2 >= 0) & (d6 <= 0)
    q = np.where(m[:, None], a + ac * t_ac[:, None], q)
    m = (vc <= 0) & (d1 >= 0) & (d3 <= 0)
    q = np.where(m[:, None], a + ab * t_ab[:, None], q)
    m = (d6 >= 0) & (d5 <= d6)
    q = np.where(m[:, None], c, q)
    m = (d3 >= 0) & (d4 <= d3)
    q = np.where(m[:, None], b, q)
    m = (d1 <= 0) & (d2 <= 0)
    q = np.where(m[:, None], a, q)
    d = p - q
    return (d * d).sum(-1)


def closest_face_indices(points: np.ndarray, mesh: TriangularMesh, k_vertices: int = 4) -> np.ndarray:
    """For each point the index of the mesh face closest in R^3 (ties -> lowest face index).

    Candidate faces are those incident to the `k_vertices` nearest mesh vertices, which contains
    the closest face of a convex, nearly regular triangulation of the sphere.
    """
    V = mesh.vertices.astype(np.float64)
    F = mesh.faces.astype(np.int64)
    P = np.asarray(points, dtype=np.float64)
    nv, nf = V.shape[0], F.shape[0]
    # vertex -> incident faces (padded table; icosphere valence is <= 6)
    flat_v = F.reshape(-1)
    flat_f = np.repeat(np.arange(nf), 3)
    order = np.argsort(flat_v, kind="stable")
    sv, sf = flat_v[order], flat_f[order]
    start = np.searchsorted(sv, np.arange(nv))
    cnt = np.bincount(sv, minlength=nv)
    maxdeg = int(cnt.max())
    table = np.full((nv, maxdeg), -1, dtype=np.int64)
    pos = np.arange(sv.shape[0]) - start[sv]
    table[sv, pos] = sf

    k = min(k_vertices, nv)
    _, near = cKDTree(V).query(P, k=k)
    near = near.reshape(P.shape[0], k)
    cand = table[near].reshape(P.shape[0], -1)  # [N, k*maxdeg], -1 = none
    best_d = np.full(P.shape[0], np.inf)
    best_f = np.full(P.shape[0], -1, dtype=np.int64)
    for j in range(cand.shape[1]):
        fj = cand[:, j]
        ok = fj >= 0
        fi = np.where(ok, fj, 0)
        d = _closest_point_sqdist(P, V[F[fi, 0]], V[F[fi, 1]], V[F[fi, 2]])
        d = np.where(ok, d, np.inf)
        better = (d < best_d) | ((d == best_d) & (fi < best_f) & ok)
        best_d = np.where(better, d, best_d)
        best_f = np.where(better, fi, best_f)
    return best_f


def in_mesh_triangle_indices(
    *, grid_latitude, grid_longitude, mesh: TriangularMesh, flat: bool = False
) -> Tuple[np.ndarray, np.ndarray]:
    """For each grid point the 3 vertices of its closest (containing) mesh face.

    Returns (grid_indices [3G] = 0,0,0,1,1,1,..., mesh_indices [3G])."""
    pts = grid_lat_lon_to_xyz(grid_latitude, grid_longitude, flat)
    face = closest_face_indices(pts, mesh)
    mesh_idx = mesh.faces[face].reshape(-1)
    grid_idx = np.repeat(np.arange(pts.shape[0]), 3)
    return grid_idx, mesh_idx


def prune_mesh_to_region(
    meshes: Sequence[TriangularMesh],
    lat_min: float, lat_max: float, lon_min: float, lon_max: float,
    buffer_deg: float = 15.0,
) -> List[TriangularMesh]:
    """Keep the finest-mesh vertices inside the lat/lon box (+buffer), the faces of every level
    whose 3 vertices all survive, re-indexed into the surviving vertex set (shared by all levels)."""
    finest = meshes[-1]
    lat, lon = get_mesh_lat_long(finest)
    lat_lo, lat_hi = max(lat_min - buffer_deg, -90.0), min(lat_max + buffer_deg, 90.0)
    lon_lo, lon_hi = lon_min - buffer_deg, lon_max + buffer_deg
    keep_lat = (lat >= lat_lo) & (lat <= lat_hi)
    if lon_lo < 0:
        keep_lon = (lon >= (lon_lo % 360)) | (lon <= lon_hi)
    elif lon_hi > 360:
        keep_lon = (lon >= lon_lo) | (lon <= (lon_hi % 360))
    else:
        keep_lon = (lon >= lon_lo) & (lon <= lon_hi)
    keep = keep_lat & keep_lon
    if not keep.any():
        raise ValueError("no mesh vertex falls inside the requested region")
    remap = np.full(keep.shape[0], -1, dtype=np.int32)
    remap[keep] = np.arange(int(keep.sum()), dtype=np.int32)
    verts = finest.vertices[keep].astype(np.float32)
    out = []
    for m in meshes:
        lvl = keep[: m.vertices.shape[0]]
        ok = lvl[m.faces].all(axis=1)
        out.append(TriangularMesh(vertices=verts, faces=remap[m.faces[ok]].astype(np.int32)))
    return out


def tile_order(points: np.ndarray, leaf: int = 64, degree: np.ndarray = None) -> np.ndarray:
    """Node order in which every run of `leaf` consecutive nodes is a spatially compact patch: recursive
    coordinate bisection along the widest axis, cut at multiples of `leaf` (ties broken by the original index,
    so the result is deterministic).  Returns `order` with `order[new] = old`.

    Not part of the reference: the mesh processor may number its rows freely because mesh nodes never leave the
    model (inputs and outputs live on grid nodes, src/models.py:808-874).  With this order the neighbours of a
    64-row tile of the mesh graph are 108 distinct rows instead of 211 in creation order, which is what lets the
    aggregation stage a tile's sources once in LDS (csrc/aggregate.hip, agg_halo_kernel).  Each row keeps the
    order of its in-edges, so every sum runs in the reference's order and results are unchanged bit for bit.
    With `degree` the nodes of a patch are listed by descending degree: the few high-degree nodes (the 642 coarse
    vertices of a [3, 5] mesh have 12 neighbours, the rest 6) then share row groups instead of each dragging three
    ordinary rows through the long-row path."""
    pts = np.asarray(points, dtype=np.float64)
    out = []
    stack = [np.arange(len(pts))]
    while stack:
        ids = stack.pop()
        if len(ids) <= leaf:
            if degree is not None:
                ids = ids[np.lexsort((ids, -np.asarray(degree)[ids]))]
            out.append(ids)
            continue
        p = pts[ids]
        ax = int(np.argmax(p.max(0) - p.min(0)))
        o = ids[np.argsort(p[:, ax], kind="stable")]
        h = (len(o) // 2 + leaf - 1) // leaf * leaf
        h = min(max(h, leaf), len(o) - 1)
        stack.append(o[h:])  # popped after the low half: keeps the low half first
        stack.append(o[:h])
    return np.concatenate(out) if out else np.zeros(0, dtype=np.int64)
