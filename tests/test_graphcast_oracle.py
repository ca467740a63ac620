"""CPU tests of the GraphCast row: the graph construction against the published node / edge counts and geometric invariants, the
oracle against its golden fixture, and the C ABI of libskyrim_graphcast.so (symbols + argument errors; no compute without a GPU)."""
import ctypes
import re
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import graphcast_graph as OG
from oracle import graphcast_oracle as O
from skyrim_amd.graphcast import engine as E
from skyrim_amd.graphcast.mesh import build_graph, edge_features, faces_to_edges, icosahedron, lat_lon_to_xyz, latitude_band, refine, shard_graph
from skyrim_amd.graphcast.spec import CHANNELS, GraphcastConfig, flops_per_step, forcings, init_synthetic, mlp_names, param_spec, synthetic_states

TINY = GraphcastConfig(n_lat=33, n_lon=64, splits=2, latent=32, steps=3)
GOLD = Path(__file__).resolve().parent / "golden" / "graphcast_tiny_33x64.npz"


def test_icosahedron_and_refinement_invariants():
    v, f = icosahedron()
    assert v.shape == (12, 3) and f.shape == (20, 3) and np.allclose(np.linalg.norm(v, axis=1), 1.0)
    e = faces_to_edges(f)
    assert len(e) == 60 and np.allclose(np.linalg.norm(v[e[:, 0]] - v[e[:, 1]], axis=1), np.linalg.norm(v[e[0, 0]] - v[e[0, 1]]))
    top = np.argsort(-v[:, 2])[:3]                                       # deepmind's orientation: a FACE on top, not a vertex
    assert np.allclose(v[top, 2], v[top[0], 2]) and sorted(top.tolist()) in [sorted(t) for t in f.tolist()]
    assert abs(np.linalg.norm(v[top].mean(0)[:2])) < 1e-12                # its centroid is the north pole
    for _ in range(3):
        v0 = v
        v, f = refine(v, f)
        assert np.array_equal(v[: len(v0)], v0)                          # coarser nodes keep their index (multi-mesh)
        assert len(v) - len(faces_to_edges(f)) // 2 + len(f) == 2         # Euler characteristic of the sphere
        assert np.allclose(np.linalg.norm(v, axis=1), 1.0)
        assert (np.einsum("ij,ij->i", np.cross(v[f[:, 1]] - v[f[:, 0]], v[f[:, 2]] - v[f[:, 0]]), v[f].sum(1)) > 0).all()   # outward faces


@pytest.mark.timeout(600)
def test_full_size_graph_has_the_published_counts():
    """Lam et al. 2023: M6 has 40 962 nodes / 81 920 faces, the multi-mesh 327 660 directed edges, and each of the 1 038 240 grid
    points receives from the 3 vertices of its triangle; ~1.6 M grid->mesh edges at 0.6 x the longest M6 edge."""
    g = build_graph(721, 1440, 6)
    assert (g.n_grid, g.n_mesh, len(g.faces), len(g.mesh_edges), len(g.m2g_edges)) == (1038240, 40962, 81920, 327660, 3114720)
    assert 1.55e6 < len(g.g2m_edges) < 1.70e6
    assert len(np.unique(g.g2m_edges[:, 1])) == g.n_mesh                  # every mesh node hears from the grid
    assert (np.diff(g.mesh_edges[:, 1]) >= 0).all() and (np.diff(g.g2m_edges[:, 1]) >= 0).all() and (np.diff(g.m2g_edges[:, 1]) >= 0).all()
    assert np.array_equal(g.m2g_edges[:, 1], np.repeat(np.arange(g.n_grid), 3))
    # barycentric check on a sample: the grid point lies inside (or on) its triangle
    idx = np.arange(0, g.n_grid, 997)
    tri = g.mesh_pos[g.m2g_edges[:, 0].reshape(-1, 3)[idx]]
    p = g.grid_pos[idx]
    for i, j in ((0, 1), (1, 2), (2, 0)):
        assert (np.einsum("ij,ij->i", np.cross(tri[:, i], tri[:, j]), p) > -1e-9).all()
    assert g.mesh_edge_feat.shape == (327660, 4) and abs(g.mesh_edge_feat[:, 0].max() - 1.0) < 1e-6 and g.grid_node_feat.shape == (1038240, 3)


def test_edge_features_are_receiver_local():
    recv = lat_lon_to_xyz(np.array([0.0, 40.0, -70.0]), np.array([0.0, 100.0, 250.0]))
    east = lat_lon_to_xyz(np.array([0.0, 40.0, -70.0]), np.array([1.0, 101.0, 251.0]))
    north = lat_lon_to_xyz(np.array([1.0, 41.0, -69.0]), np.array([0.0, 100.0, 250.0]))
    fe, fn = edge_features(east, recv), edge_features(north, recv)
    # in the receiver's frame (receiver at (1, 0, 0)): a sender to the east has +y, one to the north +z, the same at every location
    assert (fe[:, 2] > 0).all() and (np.abs(fe[:, 3]) < 0.02 * fe[:, 0]).all()
    assert (fn[:, 3] > 0).all() and (np.abs(fn[:, 2]) < 1e-6).all()
    assert np.allclose(fn[:, 0], fn[0, 0], rtol=1e-5)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_grid_sharding_partitions_the_graph(world):
    """BASELINE configs[3] / SURVEY 8(e): latitude bands of grid nodes; every grid->mesh edge belongs to the shard of its sender,
    every mesh->grid edge to the shard of its receiver; the mesh is replicated."""
    g = build_graph(33, 64, 2)
    seen_g2m, rows = 0, 0
    for r in range(world):
        lat0, lat1 = latitude_band(33, r, world)
        s = shard_graph(g, 33, 64, r, world)
        assert s.n_grid == (lat1 - lat0) * 64 and s.n_mesh == g.n_mesh and s.mesh_edges is g.mesh_edges
        assert s.g2m_edges[:, 0].min() >= 0 and s.g2m_edges[:, 0].max() < s.n_grid and (np.diff(s.g2m_edges[:, 1]) >= 0).all()
        assert np.array_equal(s.m2g_edges[:, 1], np.repeat(np.arange(s.n_grid), 3))
        assert np.array_equal(s.m2g_edges[:, 0], g.m2g_edges[3 * lat0 * 64:3 * lat1 * 64, 0])
        assert len(s.g2m_edge_feat) == len(s.g2m_edges) and len(s.m2g_edge_feat) == len(s.m2g_edges) and len(s.grid_node_feat) == s.n_grid
        seen_g2m += len(s.g2m_edges)
        rows += s.n_grid
    assert seen_g2m == len(g.g2m_edges) and rows == g.n_grid
    assert latitude_band(33, 0, world)[0] == 0 and latitude_band(33, world - 1, world)[1] == 33


def _edge_set(edges):
    return set(map(tuple, np.asarray(edges).tolist()))


def _node_map(a_pos, b_pos):
    """index in b of every node of a (the two constructions number the mesh nodes differently)."""
    from scipy.spatial import cKDTree
    d, j = cKDTree(b_pos).query(a_pos)
    assert d.max() < 1e-9 and len(set(j.tolist())) == len(a_pos)
    return j


@pytest.mark.parametrize("grid,splits", [((33, 64), 2), ((61, 120), 3), ((181, 360), 4)])
def test_product_graph_equals_the_oracles_own_construction(grid, splits):
    """mesh.py (product) against oracle/graphcast_graph.py (written separately, restating deepmind/graphcast's modules): same
    node positions up to numbering, identical edge SETS, identical features edge by edge -- so parity tests that hand the
    oracle ITS graph and the engine the product's would expose any error in either construction."""
    g = build_graph(*grid, splits)
    o = OG.build(*grid, splits)
    assert (g.n_mesh, g.n_grid) == (o.n_mesh, o.n_grid) and np.allclose(g.grid_pos, o.grid_pos, atol=1e-12)
    to_o = _node_map(g.mesh_pos, o.mesh_pos)                                          # product mesh index -> oracle mesh index
    assert _edge_set(np.stack([to_o[g.mesh_edges[:, 0]], to_o[g.mesh_edges[:, 1]]], 1)) == _edge_set(o.mesh_edges)
    assert _edge_set(np.stack([g.g2m_edges[:, 0], to_o[g.g2m_edges[:, 1]]], 1)) == _edge_set(o.g2m_edges)
    # mesh -> grid: the same triangle for EVERY grid point -- points on a mesh edge / vertex follow the shared geometric tie rule
    tg = np.sort(to_o[g.m2g_edges[:, 0]].reshape(-1, 3), axis=1)
    to_ = np.sort(o.m2g_edges[:, 0].reshape(-1, 3), axis=1)
    differ = np.nonzero((tg != to_).any(axis=1))[0]
    assert len(differ) == 0
    # features, matched edge by edge through (sender, receiver) keys

    def table(edges, feat):
        return {tuple(e): f for e, f in zip(np.asarray(edges).tolist(), feat)}

    for ge, gf, oe, of, map_s, map_r in ((g.mesh_edges, g.mesh_edge_feat, o.mesh_edges, o.mesh_edge_feat, to_o, to_o),
                                         (g.g2m_edges, g.g2m_edge_feat, o.g2m_edges, o.g2m_edge_feat, None, to_o)):
        ot = table(oe, of)
        keys = np.stack([map_s[ge[:, 0]] if map_s is not None else ge[:, 0], map_r[ge[:, 1]] if map_r is not None else ge[:, 1]], 1)
        assert np.allclose(np.stack([ot[tuple(k)] for k in keys.tolist()]), gf, atol=2e-6)
    same = np.setdiff1d(np.arange(g.n_grid), differ)
    ot = table(o.m2g_edges, o.m2g_edge_feat)
    rows = np.repeat(same, 3) * 3 + np.tile(np.arange(3), len(same))
    keys = np.stack([to_o[g.m2g_edges[rows, 0]], g.m2g_edges[rows, 1]], 1)
    assert np.allclose(np.stack([ot[tuple(k)] for k in keys.tolist()]), g.m2g_edge_feat[rows], atol=2e-6)
    assert np.allclose(g.grid_node_feat, o.grid_node_feat, atol=1e-6) and np.allclose(g.mesh_node_feat, o.mesh_node_feat[to_o], atol=1e-6)


def test_oracle_graph_follows_the_published_conventions():
    """oracle/graphcast_graph.py header: face-up icosahedron, (sin lat, cos lon, sin lon) node features, receiver-local edge
    features normalised by the longest edge, 0.6 x longest-edge radius."""
    pts, tri = OG.base_icosahedron()
    assert len(tri) == 20 and np.allclose(np.linalg.norm(pts, axis=1), 1.0)
    top = np.argsort(-pts[:, 2])[:3]
    assert np.allclose(pts[top, 2], pts[top[0], 2]) and np.allclose(pts[top].mean(0)[:2], 0.0, atol=1e-12)
    # before the rotation the top is an edge parallel to y: vertices 1 and 7 of the construction order
    raw = np.array([(0.0, 1.0, OG.PHI), (0.0, -1.0, OG.PHI)]) / np.hypot(1.0, OG.PHI)
    assert np.allclose(raw[:, 2], raw[0, 2]) and np.allclose(raw[:, 0], 0.0)
    nf = OG.node_features(OG.lat_lon_to_unit(np.array([90.0, 0.0, -30.0]), np.array([0.0, 90.0, 180.0])))
    assert np.allclose(nf, [[1.0, 1.0, 0.0], [0.0, 0.0, 1.0], [-0.5, -1.0, 0.0]], atol=1e-6)
    recv = OG.lat_lon_to_unit(np.array([0.0, 40.0, -70.0]), np.array([0.0, 100.0, 250.0]))
    east = OG.lat_lon_to_unit(np.array([0.0, 40.0, -70.0]), np.array([1.0, 101.0, 251.0]))
    north = OG.lat_lon_to_unit(np.array([1.0, 41.0, -69.0]), np.array([0.0, 100.0, 250.0]))
    fe, fn = OG.edge_features(east, recv), OG.edge_features(north, recv)
    assert (fe[:, 2] > 0).all() and (fn[:, 3] > 0).all() and (np.abs(fn[:, 2]) < 1e-6).all() and abs(fe[:, 0].max() - 1.0) < 1e-6
    o = OG.build(33, 64, 2)
    fine = OG.directed_edges(o.faces)
    longest = np.linalg.norm(o.mesh_pos[fine[:, 0]] - o.mesh_pos[fine[:, 1]], axis=1).max()
    d = np.linalg.norm(o.grid_pos[o.g2m_edges[:, 0]] - o.mesh_pos[o.g2m_edges[:, 1]], axis=1)
    assert d.max() <= 0.6 * longest and len(o.mesh_edges) == 2 * (30 + 120 + 480)      # edges of M0 + M1 + M2, both directions


def test_oracle_matches_golden_fixture():
    gold = np.load(GOLD)
    g = OG.build(TINY.n_lat, TINY.n_lon, TINY.splits)                      # the oracle's own graph
    assert np.array_equal(g.mesh_edges, gold["mesh_edges"]) and np.array_equal(g.g2m_edges[::13], gold["g2m_edges_sub"])
    assert np.array_equal(g.m2g_edges[::11, 0], gold["m2g_senders_sub"]) and np.allclose(g.mesh_edge_feat[::9], gold["mesh_edge_feat_sub"], atol=1e-6)
    p = init_synthetic(TINY, 0)
    x0, x1 = synthetic_states(TINY, 0)
    f = forcings(TINY, 1000.0)
    assert np.array_equal(x1[:, ::2, ::4].numpy(), gold["x_cur_sub"]) and np.allclose(f[:, ::4, ::8].numpy(), gold["forcing_sub"], atol=1e-6)
    taps = {}
    y = O.forward(p, g, x0, x1, f, taps=taps)
    assert (np.abs(y[:, ::2, ::4].numpy() - gold["step1_sub"]) / gold["increment_absmax"][:, None, None]).max() < 1e-4
    assert np.allclose(taps["encoder.vm"][::7].numpy(), gold["encoder_vm_sub"], atol=1e-4)
    assert np.allclose(taps["processor.vm"][::7].numpy(), gold["processor_vm_sub"], atol=1e-3)
    assert torch.equal(y, O.forward(p, g, x0, x1, f))
    # the product's graph (different node numbering and edge order) gives the same function
    y2 = O.forward(p, build_graph(TINY.n_lat, TINY.n_lon, TINY.splits), x0, x1, f)
    assert O.increment_rel_err(y2, y, x1).max().item() < 1e-4


def test_spec():
    full = GraphcastConfig()
    assert len(CHANNELS) == 83 and CHANNELS[0] == "z50" and CHANNELS[13] == "q50" and CHANNELS[-5:] == ["u10m", "v10m", "t2m", "msl", "tp06"]
    assert full.grid_in == 186 and len(mlp_names(full)) == 8 + 32 + 3
    n_param = sum(int(np.prod(s)) for n, s in param_spec(full) if n != "static")
    assert 35e6 < n_param < 37e6                                          # the paper's 36.7 M
    assert 25e12 < flops_per_step(full, 1038240, 40962, 327660, 1645760, 3114720) < 27e12
    f = forcings(TINY, 12.0)
    assert f.shape == (15, 33, 64) and float(f[0].min()) >= 0.0 and float(f[0].max()) <= 1.0 and abs(float(f[1, 0, 0]) ** 2 + float(f[2, 0, 0]) ** 2 - 1.0) < 1e-6


def test_graphcast_library_exports_declared_symbols_and_rejects_bad_arguments():
    header = (Path(__file__).resolve().parent.parent / "include" / "skyrim_graphcast.h").read_text()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    syms = sorted(set(re.findall(r"\b(skgc_[a-z0-9_]+)\s*\(", header)))
    lib = E.load_library()
    assert set(syms) == set(E.EXPORTS) and all(hasattr(lib, s) for s in syms)
    assert lib.skgc_abi_version() == 5
    assert lib.skgc_gather_gemm(None, None) == -1
    assert lib.skgc_gather_gemm(ctypes.byref(E.GatherDesc()), None) == -1
    assert lib.skgc_layer_norm(None, None, None, None, None, 4, 32, None) == -1
    assert lib.skgc_segment_sum(None, None, None, None, 4, 32, None) == -1


@pytest.mark.parametrize("n", [1, 16, 37, 2520])
def test_grouped_row_order_of_the_mesh_to_grid_kernel(n):
    """skgc_sum_desc::group = 3: virtual row 48 t + 16 a + l = edge a of node 16 t + l; every edge of every node exactly once, the
    three members of a node in the same lane position (row % 16) of three consecutive 16-row fragments."""
    from skyrim_amd.graphcast.mesh import grouped_rows_by3
    rows = grouped_rows_by3(n)
    assert len(rows) == (n + 15) // 16 * 48
    v = np.arange(len(rows))
    node = 16 * (v // 48) + v % 16
    live = node < n
    assert sorted(rows[live].tolist()) == list(range(3 * n))
    assert (rows[live] // 3 == node[live]).all() and (rows[live] % 3 == ((v % 48) // 16)[live]).all()
    assert (rows[~live] == 0).all()



def test_edge_mlp_by_distributivity_and_grouped_receiver_sum_cpu():
    """The two algebraic rewrites of the GraphCast engine, checked in float64 against the oracle's concatenated form:
    fc1(concat(e, v_s[send], v_r[recv])) = e W_e^T + (v_s W_s^T)[send] + (v_r W_r^T)[recv] + b, and -- with three edges per receiver stored
    receiver by receiver -- the receiver sum of the edge MLP's outputs taken over the kernel's virtual row order (grouped_rows_by3)."""
    from skyrim_amd.graphcast.mesh import grouped_rows_by3
    gen = torch.Generator().manual_seed(1)
    L, n_send, n_recv = 32, 11, 21
    E = 3 * n_recv
    r = lambda *s: torch.randn(*s, generator=gen, dtype=torch.float64)  # noqa: E731
    p = {"m.fc1.weight": r(L, 3 * L) / (3 * L) ** 0.5, "m.fc1.bias": r(L) * 0.1, "m.fc2.weight": r(L, L) / L ** 0.5, "m.fc2.bias": r(L) * 0.1,
         "m.ln.weight": 1 + 0.1 * r(L), "m.ln.bias": 0.1 * r(L)}
    e, vs, vr = r(E, L), r(n_send, L), r(n_recv, L)
    edges = torch.stack([torch.randint(0, n_send, (E,), generator=gen), torch.arange(n_recv).repeat_interleave(3)], dim=1)
    want_edges = O.edge_update(p, "m", e, vs, vr, edges)                           # concatenated form
    want = O.aggregate(want_edges, edges[:, 1], n_recv)
    w1 = p["m.fc1.weight"]
    we, ws, wr = w1[:, :L], w1[:, L:2 * L], w1[:, 2 * L:]
    rows = torch.from_numpy(grouped_rows_by3(n_recv))                              # virtual row -> edge
    hidden = e[rows] @ we.T + p["m.fc1.bias"] + (vs @ ws.T)[edges[rows, 0]] + (vr @ wr.T)[edges[rows, 1]]
    out = torch.nn.functional.layer_norm(torch.nn.functional.silu(hidden) @ p["m.fc2.weight"].T + p["m.fc2.bias"], (L,), p["m.ln.weight"], p["m.ln.bias"], 1e-5)
    v = torch.arange(len(rows))
    node = 16 * (v // 48) + v % 16
    got = torch.zeros(n_recv, L, dtype=torch.float64).index_add_(0, node[node < n_recv], out[node < n_recv])      # what the epilogue sums in registers
    assert (got - want).abs().max().item() < 1e-12


def test_mesh_renumbering_is_an_isomorphism_and_spatially_coherent():
    """skyrim_amd/graphcast/mesh.py: renumber_mesh(g, spatial_order(...)) -- what the engine works on since round 5 -- is the same graph (edge sets
    with their features map back one to one, edges stay sorted by receiver, the mesh->grid triples keep their grid order), and a sender's
    index is close to its receiver's (the level-by-level numbering of the multi-mesh puts them hundreds of rows apart)."""
    from skyrim_amd.graphcast.mesh import build_graph, renumber_mesh, spatial_order
    g = build_graph(61, 120, 4)
    order = spatial_order(g.mesh_pos)
    h = renumber_mesh(g, order)
    assert sorted(order.tolist()) == list(range(g.n_mesh)) and np.array_equal(h.mesh_pos, g.mesh_pos[order]) and np.array_equal(h.mesh_node_feat, g.mesh_node_feat[order])

    def rows(e, f):
        return sorted(map(tuple, np.concatenate([e, np.round(f.astype(np.float64) * 1e6).astype(np.int64)], 1).tolist()))
    me = h.mesh_edges.copy(); me[:, 0] = order[me[:, 0]]; me[:, 1] = order[me[:, 1]]
    assert rows(me, h.mesh_edge_feat) == rows(g.mesh_edges, g.mesh_edge_feat)
    g2m = h.g2m_edges.copy(); g2m[:, 1] = order[g2m[:, 1]]
    assert rows(g2m, h.g2m_edge_feat) == rows(g.g2m_edges, g.g2m_edge_feat)
    m2g = h.m2g_edges.copy(); m2g[:, 0] = order[m2g[:, 0]]
    assert np.array_equal(m2g, g.m2g_edges) and np.array_equal(h.m2g_edge_feat, g.m2g_edge_feat)
    assert (np.diff(h.mesh_edges[:, 1]) >= 0).all() and (np.diff(h.g2m_edges[:, 1]) >= 0).all()
    assert np.array_equal(np.sort(order[h.faces], axis=1), np.sort(g.faces, axis=1))
    dist = lambda e: np.median(np.abs(e[:, 0] - e[:, 1]))  # noqa: E731
    assert dist(h.mesh_edges) * 10 < dist(g.mesh_edges)
    with pytest.raises(ValueError):
        renumber_mesh(g, np.zeros(g.n_mesh, dtype=np.int64))
