"""GraphCast's graph, restated for the ORACLE independently of the product's builder (skyrim_amd/graphcast/mesh.py).

TEST INFRASTRUCTURE ONLY (imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg).  Nothing here imports
``skyrim_amd``; tests compare this construction with mesh.py edge SET by edge set and feature by feature.

PARITY UNPINNED: deepmind/graphcast (``icosahedral_mesh.py``, ``grid_mesh_connectivity.py``, ``model_utils.py``) is not in this
image; what follows restates those modules as published (Lam et al. 2023, Methods "Generating the multi-mesh / encoder / decoder
graphs", and the public source), each convention with where it comes from:

  icosahedron      12 vertices (+-1, +-phi, 0), (0, +-1, +-phi), (+-phi, 0, +-1) / |(1, phi)|, then rotated about the y axis by
                   (pi - dihedral) / 2 with dihedral = 2 asin(phi / sqrt 3), applied as ``vertices @ R_y`` -- a FACE, not a vertex,
                   ends up on top (icosahedral_mesh.get_icosahedron: "adjacent face is now top plane")
  refinement       every triangle into 4 by edge midpoints projected to the unit sphere; coarser vertices keep their indices
  multi-mesh       vertices of the finest mesh; edges = union over levels of (0->1, 1->2, 2->0) of every face = both directions
  grid -> mesh     all (grid point, mesh vertex) pairs closer than 0.6 x the longest edge of the finest mesh (Euclidean, cKDTree
                   ball query -- grid_mesh_connectivity.radius_query_indices)
  mesh -> grid     the 3 vertices of the finest-mesh triangle containing the grid point (in_mesh_triangle_indices)
  node features    (cos(colatitude) = sin(lat), cos(lon), sin(lon))  -- model_utils.get_graph_spatial_features with
                   add_node_latitude / add_node_longitude: "Using the cos of theta. From 1. (north pole) to -1 (south pole)"
                   (the paper's text says "cosine of latitude"; the code's theta is the polar angle)
  edge features    (|d|, d) / max|d| over the edge set, d = position of the sender minus position of the receiver after rotating
                   both so that the receiver sits at longitude 0, latitude 0 (Rotation.from_euler("zy", [-lon, lat]))
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
from scipy.spatial import ConvexHull, cKDTree
from scipy.spatial.transform import Rotation

PHI = (1.0 + 5.0 ** 0.5) / 2.0
RADIUS_FRACTION = 0.6


def base_icosahedron():
    pts = []
    for c1 in (1.0, -1.0):
        for c2 in (PHI, -PHI):
            pts += [(c1, c2, 0.0), (0.0, c1, c2), (c2, 0.0, c1)]
    pts = np.asarray(pts, dtype=np.float64) / np.hypot(1.0, PHI)
    dihedral = 2.0 * np.arcsin(PHI / np.sqrt(3.0))
    pts = pts @ Rotation.from_euler("y", (np.pi - dihedral) / 2.0).as_matrix()
    tri = ConvexHull(pts).simplices.astype(np.int64)            # the 20 faces; make every one counter-clockwise seen from outside
    a, b, c = pts[tri[:, 0]], pts[tri[:, 1]], pts[tri[:, 2]]
    flip = np.einsum("ij,ij->i", np.cross(b - a, c - a), a + b + c) < 0
    tri[flip] = tri[flip][:, [0, 2, 1]]
    return pts, tri


def split_faces(pts: np.ndarray, tri: np.ndarray):
    """One refinement: new vertex per undirected edge (appended after the existing ones), 4 children per face."""
    und = np.sort(np.concatenate([tri[:, [0, 1]], tri[:, [1, 2]], tri[:, [2, 0]]]), axis=1)
    uniq, inverse = np.unique(und, axis=0, return_inverse=True)
    mid = pts[uniq[:, 0]] + pts[uniq[:, 1]]
    mid /= np.linalg.norm(mid, axis=1, keepdims=True)
    new_id = len(pts) + inverse.reshape(3, len(tri))            # rows: midpoints of (0,1), (1,2), (2,0) per face
    m01, m12, m20 = new_id
    v0, v1, v2 = tri.T
    kids = np.concatenate([np.stack([v0, m01, m20], 1), np.stack([m01, v1, m12], 1), np.stack([m20, m12, v2], 1), np.stack([m01, m12, m20], 1)])
    return np.concatenate([pts, mid]), kids


def directed_edges(tri: np.ndarray) -> np.ndarray:
    """(sender, receiver) of 0->1, 1->2, 2->0 for every face."""
    return np.stack([np.concatenate([tri[:, 0], tri[:, 1], tri[:, 2]]), np.concatenate([tri[:, 1], tri[:, 2], tri[:, 0]])], axis=1)


def lat_lon_to_unit(lat_deg, lon_deg):
    theta, phi = np.deg2rad(90.0 - np.asarray(lat_deg)), np.deg2rad(np.asarray(lon_deg))          # polar angle, azimuth
    return np.stack([np.sin(theta) * np.cos(phi), np.sin(theta) * np.sin(phi), np.cos(theta)], axis=-1)


def unit_to_angles(p):
    """(azimuth phi, polar angle theta)."""
    return np.arctan2(p[..., 1], p[..., 0]), np.arccos(np.clip(p[..., 2], -1.0, 1.0))


def node_features(p: np.ndarray, angles=None) -> np.ndarray:
    """``angles`` = (azimuth, polar angle) for grid nodes (their own lat / lon; see edge_features)."""
    phi, theta = unit_to_angles(p) if angles is None else angles
    return np.stack([np.cos(theta), np.cos(phi), np.sin(phi)], axis=-1).astype(np.float32)


def edge_features(sender_pos: np.ndarray, receiver_pos: np.ndarray, receiver_angles=None, chunk: int = 1 << 20) -> np.ndarray:
    """``receiver_angles`` = (azimuth, polar angle) per edge when the receiver is a GRID point: model_utils rotates by the node's own
    (lat, lon), so the 1440 grid points sitting on a pole each keep their own longitude as local frame (positions alone cannot tell)."""
    out = np.empty((len(sender_pos), 4), dtype=np.float64)
    for i in range(0, len(sender_pos), chunk):
        s, r = sender_pos[i:i + chunk], receiver_pos[i:i + chunk]
        phi, theta = unit_to_angles(r) if receiver_angles is None else (receiver_angles[0][i:i + chunk], receiver_angles[1][i:i + chunk])
        # v' = R v with R = Ry(pi/2 - theta) Rz(-phi): the receiver goes to (1, 0, 0)
        rot = Rotation.from_euler("zy", np.stack([-phi, np.pi / 2.0 - theta], axis=1)).as_matrix()
        d = np.einsum("nij,nj->ni", rot, s) - np.einsum("nij,nj->ni", rot, r)
        out[i:i + chunk, 0] = np.linalg.norm(d, axis=1)
        out[i:i + chunk, 1:] = d
    return (out / out[:, 0].max()).astype(np.float32)


def containing_faces(points: np.ndarray, pts: np.ndarray, tri: np.ndarray) -> np.ndarray:
    """Face whose spherical triangle contains each point: candidates = faces around the nearest mesh vertices, chosen by the largest
    minimum barycentric coordinate of the point's central projection onto the face plane."""
    n_v = len(pts)
    order = np.argsort(tri.reshape(-1), kind="stable")
    owner = (order // 3)                                           # face of every (face, corner) slot, grouped by vertex
    counts = np.bincount(tri.reshape(-1), minlength=n_v)
    start = np.concatenate([[0], np.cumsum(counts)])
    width = counts.max()
    table = np.full((n_v, width), -1, dtype=np.int64)               # faces incident to every vertex (5 or 6)
    for j in range(width):
        has = counts > j
        table[has, j] = owner[start[:-1][has] + j]
    _, near = cKDTree(pts).query(points, k=min(3, n_v))
    cand = table[near].reshape(len(points), -1)                     # faces around the 3 nearest vertices
    best = np.full(len(points), -1, dtype=np.int64)
    score = np.full(len(points), -np.inf)
    best_key = np.full((len(points), 3), -np.inf)
    centroid = pts[tri].mean(axis=1)[:, ::-1]                       # (z, y, x): the tie-break key
    tol = 1e-9
    for j in range(cand.shape[1]):
        f = cand[:, j]
        ok = f >= 0
        fa = tri[np.where(ok, f, 0)]
        basis = np.stack([pts[fa[:, 0]], pts[fa[:, 1]], pts[fa[:, 2]]], axis=2)      # columns = the three vertices
        w = np.linalg.solve(basis, points[:, :, None])[:, :, 0]                    # point = sum w_i vertex_i  (w_i >= 0 inside)
        s = np.where(ok, np.minimum((w / w.sum(axis=1, keepdims=True)).min(axis=1) + tol, 0.0), -np.inf)      # 0 = contains the point (within tol)
        # a point ON an edge / vertex belongs to several faces (trimesh returns an arbitrary one): the tie goes to the face whose
        # centroid is largest in (z, y, x) order -- a rule stated on geometry, independent of how faces are numbered
        key = centroid[np.where(ok, f, 0)]
        d = key - best_key
        lex = (d[:, 0] > tol) | ((np.abs(d[:, 0]) <= tol) & ((d[:, 1] > tol) | ((np.abs(d[:, 1]) <= tol) & (d[:, 2] > tol))))
        better = (s > score) | ((s == 0.0) & (score == 0.0) & lex)
        best[better], score[better], best_key[better] = f[better], s[better], key[better]
    if (score < 0.0).any():
        raise RuntimeError("a grid point was not located in any candidate face")
    return best


@dataclass
class Graph:
    """Same field names as the product's GraphStructure so that oracle.forward reads either."""
    n_grid: int
    n_mesh: int
    mesh_pos: np.ndarray
    grid_pos: np.ndarray
    mesh_edges: np.ndarray
    g2m_edges: np.ndarray
    m2g_edges: np.ndarray
    mesh_edge_feat: np.ndarray
    g2m_edge_feat: np.ndarray
    m2g_edge_feat: np.ndarray
    mesh_node_feat: np.ndarray
    grid_node_feat: np.ndarray
    faces: np.ndarray


def build(n_lat: int, n_lon: int, splits: int) -> Graph:
    pts, tri = base_icosahedron()
    level_faces = [tri]
    for _ in range(splits):
        pts, tri = split_faces(pts, tri)
        level_faces.append(tri)
    mesh_edges = np.unique(np.concatenate([directed_edges(f) for f in level_faces]), axis=0)
    lat = np.linspace(90.0, -90.0, n_lat)                        # the reference's grid: 90 .. -90, 0 .. 360 (pangu.py:33-34)
    lon = np.arange(n_lon) * (360.0 / n_lon)
    grid_lat, grid_lon = np.repeat(lat, n_lon), np.tile(lon, n_lat)
    grid = lat_lon_to_unit(grid_lat, grid_lon)
    fine = directed_edges(tri)
    longest = np.linalg.norm(pts[fine[:, 0]] - pts[fine[:, 1]], axis=1).max()
    hits = cKDTree(pts).query_ball_point(grid, r=RADIUS_FRACTION * longest)
    g2m = np.array([(g, m) for g, ms in enumerate(hits) for m in ms], dtype=np.int64).reshape(-1, 2)
    face = containing_faces(grid, pts, tri)
    m2g = np.stack([tri[face].reshape(-1), np.repeat(np.arange(len(grid)), 3)], axis=1)
    return Graph(n_grid=len(grid), n_mesh=len(pts), mesh_pos=pts, grid_pos=grid, mesh_edges=mesh_edges, g2m_edges=g2m, m2g_edges=m2g,
                 mesh_edge_feat=edge_features(pts[mesh_edges[:, 0]], pts[mesh_edges[:, 1]]),
                 g2m_edge_feat=edge_features(grid[g2m[:, 0]], pts[g2m[:, 1]]),
                 m2g_edge_feat=edge_features(pts[m2g[:, 0]], grid[m2g[:, 1]],
                                             (np.deg2rad(grid_lon[m2g[:, 1]]), np.deg2rad(90.0 - grid_lat[m2g[:, 1]]))),
                 mesh_node_feat=node_features(pts),
                 grid_node_feat=node_features(grid, (np.deg2rad(grid_lon), np.deg2rad(90.0 - grid_lat))), faces=tri)
