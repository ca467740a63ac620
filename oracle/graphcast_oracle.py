"""CPU restatement of one GraphCast step.  TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, never by the product path (skyrim_amd/).

PARITY UNPINNED.  The reference runs DeepMind's JAX GraphCast through earth2mip
(/root/reference/skyrim/core/models/graphcast.py:51-54, 102-118); jax, the graphcast package and the checkpoint are not
available here and the reference's tests hold no numerical vector for it.  This file restates the paper's
Methods (Lam et al. 2023): grid-node / mesh-node / edge embedders, a grid->mesh interaction network, 16 interaction
networks on the multi-mesh, a mesh->grid interaction network, an output MLP predicting the normalised residual of the
latest state; every MLP = Linear -> swish -> Linear (-> LayerNorm), latent 512, sum aggregation at the receiver,
residual updates of nodes and edges.  The graph (multi-mesh, grid<->mesh edges, structural features) has its OWN restatement in
oracle/graphcast_graph.py (``build``); parity tests hand ``forward`` that graph, not the product's (skyrim_amd/graphcast/mesh.py is
compared with it edge set by edge set in tests/test_graphcast_oracle.py).  Nothing under oracle/ imports skyrim_amd.

What deepmind/graphcast's typed-graph network has beyond what is computed here (none of it can change the output): the
mesh->grid GNN also updates its MESH nodes (an MLP 512 -> 512 -> 512 + LayerNorm whose result is never read: 0.53 M parameters),
and the grid->mesh embedder pads the mesh nodes' 3 structural features with zeros to the grid-node input width (zero inputs:
+0.24 M weights at 474 inputs).  Together with the 37-level model's wider input / output layers (474 / 227 vs 186 / 83: 0.22 M)
they account for 1.0 M of the 1.3 M between this network's 35.4 M parameters and the paper's 36.7 M; 0.35 M (1 %) are not
identified.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F


def processor_steps(p: dict) -> int:
    return 1 + max(int(k.split(".")[1]) for k in p if k.startswith("proc."))


def mlp(p: dict, name: str, x: torch.Tensor) -> torch.Tensor:
    h = F.silu(F.linear(x, p[name + ".fc1.weight"], p[name + ".fc1.bias"]))
    y = F.linear(h, p[name + ".fc2.weight"], p[name + ".fc2.bias"])
    if name + ".ln.weight" in p:
        y = F.layer_norm(y, (y.shape[-1],), p[name + ".ln.weight"], p[name + ".ln.bias"], 1e-5)
    return y


def edge_update(p: dict, name: str, e: torch.Tensor, v_send: torch.Tensor, v_recv: torch.Tensor, edges: torch.Tensor, chunk: int = 1 << 18) -> torch.Tensor:
    """MLP(concat(edge latent, sender latent, receiver latent)), in row chunks so that the concatenated rows of the 3.1 M
    mesh->grid edges (19 GB at once) never exist as one tensor."""
    out = []
    for i in range(0, len(edges), chunk):
        s = slice(i, i + chunk)
        out.append(mlp(p, name, torch.cat([e[s], v_send[edges[s, 0]], v_recv[edges[s, 1]]], dim=1)))
    return torch.cat(out)


def aggregate(e: torch.Tensor, receivers: torch.Tensor, n: int) -> torch.Tensor:
    return torch.zeros(n, e.shape[1], dtype=e.dtype).index_add_(0, receivers, e)


def forward(p: dict, graph, x_prev: torch.Tensor, x_cur: torch.Tensor, forcing: torch.Tensor, cfg=None, taps: dict | None = None):
    """(n_vars, n_lat, n_lon) x 2 + (15, n_lat, n_lon) forcings -> next state (n_vars, n_lat, n_lon).  ``graph``: graphcast_graph.build(...)
    (any object with its fields; edge order is free).  ``cfg`` is accepted for call-site symmetry with the engine and not read: the
    number of processor steps is what the parameter dict holds."""
    t = lambda a: torch.from_numpy(a)  # noqa: E731
    mean, std = p["norm.mean"][:, None, None], p["norm.std"][:, None, None]
    feats = torch.cat([(x_prev - mean) / std, (x_cur - mean) / std, forcing, p["static"]], dim=0).flatten(1).T      # [n_grid][2V + 17]
    vg = mlp(p, "embed.grid", torch.cat([feats, t(graph.grid_node_feat)], dim=1))
    vm = mlp(p, "embed.mesh", t(graph.mesh_node_feat))
    e1 = mlp(p, "embed.g2m_edge", t(graph.g2m_edge_feat))
    em = mlp(p, "embed.mesh_edge", t(graph.mesh_edge_feat))
    e2 = mlp(p, "embed.m2g_edge", t(graph.m2g_edge_feat))
    g2m, me, m2g = t(graph.g2m_edges), t(graph.mesh_edges), t(graph.m2g_edges)
    # encoder: grid -> mesh
    e1 = edge_update(p, "g2m.edge", e1, vg, vm, g2m)
    vm = vm + mlp(p, "g2m.mesh_node", torch.cat([vm, aggregate(e1, g2m[:, 1], graph.n_mesh)], dim=1))
    vg = vg + mlp(p, "g2m.grid_node", vg)
    if taps is not None:
        taps["encoder.vm"], taps["encoder.vg"] = vm, vg
    # processor
    for i in range(processor_steps(p)):
        de = edge_update(p, f"proc.{i}.edge", em, vm, vm, me)
        vm = vm + mlp(p, f"proc.{i}.node", torch.cat([vm, aggregate(de, me[:, 1], graph.n_mesh)], dim=1))
        em = em + de
    if taps is not None:
        taps["processor.vm"] = vm
    # decoder: mesh -> grid
    e2 = edge_update(p, "m2g.edge", e2, vm, vg, m2g)
    vg = vg + mlp(p, "m2g.grid_node", torch.cat([vg, aggregate(e2, m2g[:, 1], graph.n_grid)], dim=1))
    out = mlp(p, "out", vg)                                                       # [n_grid][n_vars], normalised residual
    return x_cur + (out * p["norm.diff_std"][None, :]).T.reshape(x_cur.shape)


def per_channel_rel_err(y: torch.Tensor, ref: torch.Tensor) -> torch.Tensor:
    y, ref = y.double(), ref.double()
    return (y - ref).abs().amax(dim=(-2, -1)) / ref.abs().amax(dim=(-2, -1)).clamp_min(1e-30)


def increment_rel_err(y: torch.Tensor, ref: torch.Tensor, x_cur: torch.Tensor) -> torch.Tensor:
    """Error relative to the size of the predicted CHANGE (the network's actual output), per channel."""
    d = (ref - x_cur).double()
    return (y.double() - ref.double()).abs().amax(dim=(-2, -1)) / d.abs().amax(dim=(-2, -1)).clamp_min(1e-30)
