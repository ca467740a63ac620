"""Graph structure of GraphCast (Lam et al. 2023, and deepmind/graphcast's icosahedral_mesh.py / grid_mesh_connectivity.py):

* the icosahedral multi-mesh: a regular icosahedron refined ``splits`` times by edge bisection (M0 ... M_splits; the nodes of a
  coarser mesh are a subset of the finer one's), its edge set the UNION of the edges of all levels, both directions
  (splits = 6: 40 962 nodes, 81 920 finest faces, 327 660 directed edges);
* grid -> mesh edges: every lat/lon grid point sends to the finest-mesh nodes within 0.6 x (longest finest-mesh edge);
* mesh -> grid edges: every grid point receives from the 3 vertices of the finest-mesh triangle that contains it;
* structural features: nodes (sin lat, cos lon, sin lon); edges (length, and the sender's position relative to the receiver in
  the receiver's local frame -- rotated so that the receiver sits at lat = lon = 0), scaled by the longest edge of the set.

Everything here is host-side preparation (numpy / scipy.spatial), done once per geometry.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
from scipy.spatial import cKDTree


def icosahedron():
    phi = (1.0 + np.sqrt(5.0)) / 2.0
    v = []
    for a in (1.0, -1.0):
        for b in (phi, -phi):
            v += [(a, b, 0.0), (0.0, a, b), (b, 0.0, a)]
    v = np.array(v, dtype=np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    # faces = all vertex triples at mutual distance = edge length
    d = np.linalg.norm(v[:, None] - v[None], axis=-1)
    edge = d[d > 1e-9].min()
    adj = np.abs(d - edge) < 1e-6
    faces = []
    for i in range(12):
        for j in range(i + 1, 12):
            if not adj[i, j]:
                continue
            for k in range(j + 1, 12):
                if adj[i, k] and adj[j, k]:
                    f = [i, j, k]
                    if np.dot(np.cross(v[j] - v[i], v[k] - v[i]), v[i] + v[j] + v[k]) < 0:      # outward orientation
                        f = [i, k, j]
                    faces.append(f)
    # orientation of deepmind/graphcast's icosahedral_mesh.get_icosahedron: out of the box the top is an edge parallel to y; a turn
    # about y by half the supplement of the dihedral angle lays one of its two faces flat on top (a FACE faces the north pole)
    turn = (np.pi - 2.0 * np.arcsin(phi / np.sqrt(3.0))) / 2.0
    c, s = np.cos(turn), np.sin(turn)
    v = v @ np.array([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]])
    return v, np.array(faces, dtype=np.int64)


def refine(vertices: np.ndarray, faces: np.ndarray):
    """Split every triangle in 4 by bisecting its edges (midpoints projected to the sphere); existing vertices keep their index."""
    verts = list(vertices)
    cache: dict = {}

    def mid(a, b):
        key = (a, b) if a < b else (b, a)
        if key not in cache:
            p = verts[a] + verts[b]
            verts.append(p / np.linalg.norm(p))
            cache[key] = len(verts) - 1
        return cache[key]

    out = []
    for a, b, c in faces:
        ab, bc, ca = mid(a, b), mid(b, c), mid(c, a)
        out += [[a, ab, ca], [ab, b, bc], [ca, bc, c], [ab, bc, ca]]
    return np.array(verts), np.array(out, dtype=np.int64)


def faces_to_edges(faces: np.ndarray) -> np.ndarray:
    """Directed edges (sender, receiver) of a triangle mesh, both directions, unique."""
    e = np.concatenate([faces[:, [0, 1]], faces[:, [1, 2]], faces[:, [2, 0]]])
    e = np.concatenate([e, e[:, ::-1]])
    return np.unique(e, axis=0)


def lat_lon_to_xyz(lat_deg, lon_deg):
    lat, lon = np.deg2rad(lat_deg), np.deg2rad(lon_deg)
    return np.stack([np.cos(lat) * np.cos(lon), np.cos(lat) * np.sin(lon), np.sin(lat)], axis=-1)


def xyz_to_lat_lon(p):
    return np.arcsin(np.clip(p[..., 2], -1.0, 1.0)), np.arctan2(p[..., 1], p[..., 0])


def edge_features(pos_send: np.ndarray, pos_recv: np.ndarray, recv_lat_lon=None) -> np.ndarray:
    """(length, dx, dy, dz of the sender relative to the receiver in the receiver's local frame) / longest length.
    ``recv_lat_lon`` (radians): the receiver's own latitude / longitude when it is a grid point -- the grid points on a pole share
    one position but keep their own longitude as local frame (deepmind/graphcast rotates by the node's lat / lon)."""
    lat, lon = xyz_to_lat_lon(pos_recv) if recv_lat_lon is None else recv_lat_lon
    # rotate about z by -lon, then about y so that the receiver goes to (1, 0, 0)
    cl, sl, cp, sp = np.cos(lon), np.sin(lon), np.cos(lat), np.sin(lat)

    def rot(p):
        x1 = cl * p[:, 0] + sl * p[:, 1]
        y1 = -sl * p[:, 0] + cl * p[:, 1]
        z1 = p[:, 2]
        return np.stack([cp * x1 + sp * z1, y1, -sp * x1 + cp * z1], axis=-1)

    rel = rot(pos_send) - rot(pos_recv)
    length = np.linalg.norm(rel, axis=-1, keepdims=True)
    f = np.concatenate([length, rel], axis=-1)
    return (f / length.max()).astype(np.float32)


def node_features(pos: np.ndarray, lat_lon=None) -> np.ndarray:
    """(sin lat, cos lon, sin lon): deepmind/graphcast's model_utils takes the cosine of the POLAR angle ("from 1 at the north pole
    to -1 at the south pole") and then cos / sin of the longitude."""
    lat, lon = xyz_to_lat_lon(pos) if lat_lon is None else lat_lon
    return np.stack([np.sin(lat), np.cos(lon), np.sin(lon)], axis=-1).astype(np.float32)


@dataclass
class GraphStructure:
    n_grid: int
    n_mesh: int
    mesh_pos: np.ndarray         # [n_mesh][3]
    grid_pos: np.ndarray         # [n_grid][3]
    mesh_edges: np.ndarray       # [E_mesh][2] (sender, receiver), multi-mesh, sorted by receiver
    g2m_edges: np.ndarray        # [E_g2m][2]  (grid sender, mesh receiver), sorted by receiver
    m2g_edges: np.ndarray        # [3 n_grid][2] (mesh sender, grid receiver), sorted by receiver
    mesh_edge_feat: np.ndarray   # [E][4]
    g2m_edge_feat: np.ndarray
    m2g_edge_feat: np.ndarray
    mesh_node_feat: np.ndarray   # [n_mesh][3]
    grid_node_feat: np.ndarray   # [n_grid][3]
    faces: np.ndarray            # finest faces


def _sort_by_receiver(edges: np.ndarray) -> np.ndarray:
    order = np.lexsort((edges[:, 0], edges[:, 1]))
    return edges[order]


def containing_triangles(points: np.ndarray, verts: np.ndarray, faces: np.ndarray) -> np.ndarray:
    """Index of the spherical triangle containing each unit vector.  A point lying ON an edge or a vertex belongs to several
    triangles; the tie goes to the one whose centroid is largest in (z, y, x) order (a geometric rule, so that any two
    constructions of the same mesh agree whatever their face numbering)."""
    cent = verts[faces].mean(axis=1)
    cent /= np.linalg.norm(cent, axis=1, keepdims=True)
    tree = cKDTree(cent)
    k = min(12, len(faces))
    _, cand = tree.query(points, k=k)
    a, b, c = verts[faces[:, 0]], verts[faces[:, 1]], verts[faces[:, 2]]
    nab, nbc, nca = np.cross(a, b), np.cross(b, c), np.cross(c, a)          # inward-pointing for outward-oriented faces
    best = np.full(len(points), -1, dtype=np.int64)
    best_margin = np.full(len(points), -np.inf)
    best_key = np.full((len(points), 3), -np.inf)
    tol = 1e-9
    for j in range(k):
        f = cand[:, j]
        d1, d2, d3 = np.einsum("ij,ij->i", nab[f], points), np.einsum("ij,ij->i", nbc[f], points), np.einsum("ij,ij->i", nca[f], points)
        m = np.minimum(np.minimum(d1, d2), d3) / (d1 + d2 + d3)   # smallest barycentric weight of the point's central projection
        m = np.minimum(m + tol, 0.0)                             # every containing triangle (within tol) counts the same; only "outside" is worse
        d = cent[f][:, ::-1] - best_key
        lex = (d[:, 0] > tol) | ((np.abs(d[:, 0]) <= tol) & ((d[:, 1] > tol) | ((np.abs(d[:, 1]) <= tol) & (d[:, 2] > tol))))
        upd = (m > best_margin) | ((m == 0.0) & (best_margin == 0.0) & lex)
        best[upd], best_margin[upd], best_key[upd] = f[upd], m[upd], cent[f][:, ::-1][upd]
    if (best_margin < 0.0).any():
        raise RuntimeError("a grid point lies in none of its candidate triangles")
    return best


def build_graph(n_lat: int, n_lon: int, splits: int) -> GraphStructure:
    v, f = icosahedron()
    levels = [(v, f)]
    for _ in range(splits):
        v, f = refine(v, f)
        levels.append((v, f))
    mesh_edges = np.unique(np.concatenate([faces_to_edges(ff) for _, ff in levels]), axis=0)
    mesh_edges = _sort_by_receiver(mesh_edges)
    lat = np.linspace(90.0, -90.0, n_lat)
    lon = np.arange(n_lon) * (360.0 / n_lon)
    glat, glon = np.repeat(lat, n_lon), np.tile(lon, n_lat)
    grid_pos = lat_lon_to_xyz(glat, glon)
    fine_edges = faces_to_edges(f)
    longest = np.linalg.norm(v[fine_edges[:, 0]] - v[fine_edges[:, 1]], axis=1).max()
    tree = cKDTree(v)
    near = tree.query_ball_point(grid_pos, r=0.6 * longest)
    send = np.repeat(np.arange(len(grid_pos)), [len(x) for x in near])
    recv = np.concatenate([np.asarray(x, dtype=np.int64) for x in near])
    g2m = _sort_by_receiver(np.stack([send, recv], axis=1))
    tri = containing_triangles(grid_pos, v, f)
    m2g = np.stack([f[tri].reshape(-1), np.repeat(np.arange(len(grid_pos)), 3)], axis=1)       # already sorted by receiver
    return GraphStructure(
        n_grid=len(grid_pos), n_mesh=len(v), mesh_pos=v, grid_pos=grid_pos, mesh_edges=mesh_edges, g2m_edges=g2m, m2g_edges=m2g,
        mesh_edge_feat=edge_features(v[mesh_edges[:, 0]], v[mesh_edges[:, 1]]),
        g2m_edge_feat=edge_features(grid_pos[g2m[:, 0]], v[g2m[:, 1]]),
        m2g_edge_feat=edge_features(v[m2g[:, 0]], grid_pos[m2g[:, 1]], (np.deg2rad(glat[m2g[:, 1]]), np.deg2rad(glon[m2g[:, 1]]))),
        mesh_node_feat=node_features(v), grid_node_feat=node_features(grid_pos, (np.deg2rad(glat), np.deg2rad(glon))), faces=f)


def spatial_order(pos: np.ndarray, bits: int = 10) -> np.ndarray:
    """Node indices in a locality-preserving order: sorted by the Morton (Z-order) code of the unit vectors quantised to ``bits`` bits per
    axis -- neighbours on the sphere get neighbouring positions most of the time, at every scale of the multi-mesh."""
    q = np.clip(((pos + 1.0) * 0.5 * ((1 << bits) - 1) + 0.5).astype(np.int64), 0, (1 << bits) - 1)
    code = np.zeros(len(pos), dtype=np.int64)
    for b in range(bits):
        for axis in range(3):
            code |= ((q[:, axis] >> b) & 1) << (3 * b + axis)
    return np.argsort(code, kind="stable")


def renumber_mesh(g: GraphStructure, order: np.ndarray) -> GraphStructure:
    """The same graph with its MESH nodes renumbered: new node k is old node ``order[k]``.  Mesh nodes carry latents only (the network's
    inputs and outputs live on the grid), so this changes nothing a caller can see except the order in which a receiver's messages are
    summed; edges are re-sorted by (receiver, sender) in the new numbering and their features travel with them."""
    order = np.asarray(order, dtype=np.int64)
    if sorted(order.tolist()) != list(range(g.n_mesh)):
        raise ValueError("renumber_mesh: `order` must be a permutation of the mesh nodes")
    new = np.empty(g.n_mesh, dtype=np.int64)
    new[order] = np.arange(g.n_mesh)

    def resort(edges, feat, send_is_mesh, recv_is_mesh):
        e = edges.copy()
        if send_is_mesh:
            e[:, 0] = new[e[:, 0]]
        if recv_is_mesh:
            e[:, 1] = new[e[:, 1]]
        k = np.lexsort((e[:, 0], e[:, 1])) if recv_is_mesh else np.arange(len(e))    # grid receivers keep their order (3 edges per grid node)
        return e[k], feat[k]

    me, mef = resort(g.mesh_edges, g.mesh_edge_feat, True, True)
    g2m, g2mf = resort(g.g2m_edges, g.g2m_edge_feat, False, True)
    m2g, m2gf = resort(g.m2g_edges, g.m2g_edge_feat, True, False)
    return GraphStructure(n_grid=g.n_grid, n_mesh=g.n_mesh, mesh_pos=g.mesh_pos[order], grid_pos=g.grid_pos, mesh_edges=me, g2m_edges=g2m, m2g_edges=m2g,
                          mesh_edge_feat=mef, g2m_edge_feat=g2mf, m2g_edge_feat=m2gf, mesh_node_feat=g.mesh_node_feat[order],
                          grid_node_feat=g.grid_node_feat, faces=new[g.faces])


def latitude_band(n_lat: int, rank: int, world: int) -> tuple[int, int]:
    """Rows [lat0, lat1) of the grid owned by ``rank`` (contiguous bands, sizes differ by at most one row)."""
    return rank * n_lat // world, (rank + 1) * n_lat // world


def shard_graph(g: GraphStructure, n_lat: int, n_lon: int, rank: int, world: int) -> GraphStructure:
    """The part of the graph a rank of a grid-sharded run works on: its latitude band of grid nodes, the grid->mesh edges SENT
    by them and the mesh->grid edges RECEIVED by them; the mesh (nodes, multi-mesh edges) is replicated.  The only exchange of a
    step is the sum over ranks of the per-mesh-node aggregate of the grid->mesh messages (n_mesh x latent floats)."""
    lat0, lat1 = latitude_band(n_lat, rank, world)
    p0, p1 = lat0 * n_lon, lat1 * n_lon
    keep = (g.g2m_edges[:, 0] >= p0) & (g.g2m_edges[:, 0] < p1)
    g2m = g.g2m_edges[keep].copy()
    g2m[:, 0] -= p0
    m2g = g.m2g_edges[3 * p0:3 * p1].copy()                      # sorted by receiver: a contiguous slice
    m2g[:, 1] -= p0
    return GraphStructure(
        n_grid=p1 - p0, n_mesh=g.n_mesh, mesh_pos=g.mesh_pos, grid_pos=g.grid_pos[p0:p1], mesh_edges=g.mesh_edges, g2m_edges=g2m, m2g_edges=m2g,
        mesh_edge_feat=g.mesh_edge_feat, g2m_edge_feat=g.g2m_edge_feat[keep], m2g_edge_feat=g.m2g_edge_feat[3 * p0:3 * p1],
        mesh_node_feat=g.mesh_node_feat, grid_node_feat=g.grid_node_feat[p0:p1], faces=g.faces)


def grouped_rows_by3(n_groups: int) -> np.ndarray:
    """Row order of the mesh->grid edge kernel that sums a grid node's three edges in its epilogue (skgc_sum_desc::group = 3,
    include/skyrim_graphcast.h): virtual row 48 t + 16 a + l is edge a of node 16 t + l, i.e. edge index 3 (16 t + l) + a when the
    edges are stored node by node; rows of nodes >= n_groups (the ragged last 16) point at edge 0 and are never stored."""
    v = np.arange((n_groups + 15) // 16 * 48)
    node = 16 * (v // 48) + v % 16
    return np.where(node < n_groups, 3 * np.minimum(node, n_groups - 1) + (v % 48) // 16, 0).astype(np.int64)

