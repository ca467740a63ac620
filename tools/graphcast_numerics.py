"""Numerics plan of the round-4 GraphCast kernels, emulated on the CPU oracle (production width and depth, reduced grid / mesh).

Every place where the fused kernels narrow a value is a switch here; the reference is the oracle in float64.  "f16(x)" = round to
ONE fp16 plane; a GEMM whose activation operand is one fp16 plane and whose weights are fp16 hi/lo planes (two MFMA terms) is emulated
as linear(f16(x), W) -- the dropped W_lo * x_lo term is 2^-22 relative.

    python tools/graphcast_numerics.py [n_lat n_lon splits latent steps]

Switches (comma list in GCNUM_PLAN, default = the shipped plan):
    edge_store   processor edge latents stored as one fp16 plane (the residual update rounds once per layer)
    edge_hidden  hidden activation of the edge MLPs (operand of fc2) one fp16 plane
    static16     input-independent first-Linear terms of the encoder / decoder edge MLPs stored as one fp16 plane
    nodeterm16   node terms v W_s^T / v W_r^T stored as one fp16 plane
    nodeterm_a   node-term GEMMs take f16(v) (two MFMA terms) instead of the hi/lo pair
    node_a       node MLPs' fc1 operand (concat(v, agg)) one fp16 plane
    node_hidden  node MLPs' hidden activation one fp16 plane
    embed_a      grid embedder's input features one fp16 plane (after normalisation)
    out_a        output MLP's operand / hidden one fp16 plane
    edge_w1_16 / edge_w2_16   only W_e of the processor / only the second Linear of the edge MLPs as one fp16 plane
    edge_w16     weights of the edge MLPs' GEMMs (W_e of the processor, fc2 everywhere) one fp16 plane (ONE MFMA term with edge_store / edge_hidden)
    gridnode_w16 / gridnode_a / gridnode_hidden   the two GRID-node MLPs of the encoder / decoder only (1 M rows each at full size: 12 of the
                 step's 55 ms): weights as one fp16 plane (two MFMA terms, half the bytes every workgroup streams) / fc1 operand / hidden one plane
    meshnode_w16 the mesh-node MLPs (encoder + the 16 processor layers): weights as one fp16 plane
"""
import os
import sys
from pathlib import Path

import torch
import torch.nn.functional as F

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle import graphcast_graph as OG  # noqa: E402
from oracle import graphcast_oracle as O  # noqa: E402
from skyrim_amd.graphcast.spec import GraphcastConfig, forcings, init_synthetic, synthetic_states  # noqa: E402


def f16(x):
    return x.to(torch.float16).to(x.dtype)


def forward(p, graph, x_prev, x_cur, forcing, plan, taps=None):
    t = lambda a: torch.from_numpy(a).double()  # noqa: E731
    on = lambda k: k in plan  # noqa: E731
    rd = lambda k, x: f16(x) if on(k) else x  # noqa: E731
    L = p["embed.mesh.fc2.weight"].shape[0]

    def ln(name, y):
        return F.layer_norm(y, (y.shape[-1],), p[name + ".ln.weight"], p[name + ".ln.bias"], 1e-5) if name + ".ln.weight" in p else y

    def mlp(name, x, ka=None, kh=None, kw=None, ka2=None, kh2=None):
        w = lambda m: rd(kw, m) if kw else m  # noqa: E731
        a = rd(ka2, x) if ka2 else x
        h = F.silu(F.linear(rd(ka, a) if ka else a, w(p[name + ".fc1.weight"]), p[name + ".fc1.bias"]))
        h = rd(kh2, h) if kh2 else h
        if taps is not None:
            taps[name + ".fc1"], taps[name + ".fc2"] = x, h
        return ln(name, F.linear(rd(kh, h) if kh else h, w(p[name + ".fc2.weight"]), p[name + ".fc2.bias"]))

    def edge_mlp(name, e_term, vs, s_idx, vr, r_idx):
        """e_term: the (possibly stored) e W_e^T + b1 (+ folded receiver term); node terms by distributivity."""
        w1 = p[name + ".fc1.weight"]
        pre = e_term
        if vs is not None:
            pre = pre + rd("nodeterm16", F.linear(rd("nodeterm_a", vs), w1[:, L:2 * L]))[s_idx]
        if vr is not None:
            pre = pre + rd("nodeterm16", F.linear(rd("nodeterm_a", vr), w1[:, 2 * L:]))[r_idx]
        h = rd("edge_hidden", F.silu(pre))
        return ln(name, F.linear(h, rd("edge_w16", rd("edge_w2_16", p[name + ".fc2.weight"])), p[name + ".fc2.bias"]))

    mean, std = p["norm.mean"][:, None, None], p["norm.std"][:, None, None]
    feats = torch.cat([(x_prev - mean) / std, (x_cur - mean) / std, forcing, p["static"]], dim=0).flatten(1).T
    vg = mlp("embed.grid", torch.cat([feats, t(graph.grid_node_feat)], dim=1), "embed_a", "node_hidden")
    vm = mlp("embed.mesh", t(graph.mesh_node_feat))
    e1 = mlp("embed.g2m_edge", t(graph.g2m_edge_feat))
    em = mlp("embed.mesh_edge", t(graph.mesh_edge_feat))
    e2 = mlp("embed.m2g_edge", t(graph.m2g_edge_feat))
    g2m, me, m2g = (torch.from_numpy(a) for a in (graph.g2m_edges, graph.mesh_edges, graph.m2g_edges))
    agg = lambda e, r, n: torch.zeros(n, L, dtype=e.dtype).index_add_(0, r, e)  # noqa: E731
    # encoder: static = e1 W_e^T + b1 + (vm0 W_r^T)[recv], prepared once
    w1 = p["g2m.edge.fc1.weight"]
    st1 = rd("static16", F.linear(e1, w1[:, :L], p["g2m.edge.fc1.bias"]) + F.linear(vm, w1[:, 2 * L:])[g2m[:, 1]])
    y1 = edge_mlp("g2m.edge", st1, vg, g2m[:, 0], None, None)
    vm = vm + mlp("g2m.mesh_node", torch.cat([vm, agg(y1, g2m[:, 1], graph.n_mesh)], dim=1), "node_a", "node_hidden", "meshnode_w16")
    vg = vg + mlp("g2m.grid_node", vg, "node_a", "node_hidden", "gridnode_w16", "gridnode_a", "gridnode_hidden")
    em = rd("edge_store", em)
    for i in range(O.processor_steps(p)):
        name = f"proc.{i}.edge"
        w1 = p[name + ".fc1.weight"]
        de = edge_mlp(name, F.linear(em, rd("edge_w16", rd("edge_w1_16", w1[:, :L])), p[name + ".fc1.bias"]), vm, me[:, 0], vm, me[:, 1])
        vm = vm + mlp(f"proc.{i}.node", torch.cat([vm, agg(de, me[:, 1], graph.n_mesh)], dim=1), "node_a", "node_hidden", "meshnode_w16")
        em = rd("edge_store", em + de)
    w1 = p["m2g.edge.fc1.weight"]
    st2 = rd("static16", F.linear(e2, w1[:, :L], p["m2g.edge.fc1.bias"]))
    y2 = edge_mlp("m2g.edge", st2, vm, m2g[:, 0], vg, m2g[:, 1])
    vg = vg + mlp("m2g.grid_node", torch.cat([vg, agg(y2, m2g[:, 1], graph.n_grid)], dim=1), "node_a", "node_hidden", "gridnode_w16", "gridnode_a", "gridnode_hidden")
    out = mlp("out", vg, "out_a", "out_a")
    return x_cur + (out * p["norm.diff_std"][None, :]).T.reshape(x_cur.shape)


def main():
    a = [int(v) for v in sys.argv[1:6]]
    n_lat, n_lon, splits, latent, steps = a + [61, 120, 4, 512, 16][len(a):]
    cfg = GraphcastConfig(n_lat=n_lat, n_lon=n_lon, splits=splits, latent=latent, steps=steps)
    p = {k: v.double() for k, v in init_synthetic(cfg, 0).items()}
    og = OG.build(cfg.n_lat, cfg.n_lon, cfg.splits)
    x0, x1 = (x.double() for x in synthetic_states(cfg, 0))
    fk = forcings(cfg, 1000.0).double()
    ref = forward(p, og, x0, x1, fk, set())
    chk = O.forward({k: v.float() for k, v in p.items()}, og, x0.float(), x1.float(), fk.float()).double()      # the oracle proper (fp32)
    print(f"grid {n_lat}x{n_lon} M{splits} latent {latent} steps {steps}; emulation with no switch vs oracle: {O.increment_rel_err(ref, chk, x1).max().item():.2e}")
    if os.environ.get("GCNUM_COMPENSATED"):
        # the weights of the named MLPs as ONE fp16 plane, rounded with error feedback against the operand statistics of ANOTHER state pair
        # (skyrim_amd/pangu/calibration.py: compensated_round + bias fold), nearest rounding beside it
        from skyrim_amd.pangu.calibration import compensated_round, operand_statistics
        names = os.environ["GCNUM_COMPENSATED"].split(",")
        c0, c1 = (x.double() for x in synthetic_states(cfg, 1))
        taps = {}
        forward(p, og, c0, c1, forcings(cfg, 2000.0).double(), set(), taps)
        for rounding in ("nearest", "nearest + bias fold", "compensated"):
            q = dict(p)
            for name in names:
                for fc in ("fc1", "fc2"):
                    w, b, x = p[f"{name}.{fc}.weight"], p[f"{name}.{fc}.bias"], taps[f"{name}.{fc}"]
                    mu, cov = operand_statistics(x.float())
                    wq = compensated_round(w.float(), cov + torch.outer(mu, mu)).double() if rounding == "compensated" else f16(w)
                    q[f"{name}.{fc}.weight"] = wq
                    if rounding != "nearest":
                        q[f"{name}.{fc}.bias"] = b + (w - wq) @ mu
            for label, plan in (("alone", set()), ("with the shipped plan", {"edge_store", "edge_hidden", "static16", "edge_w1_16"})):
                y = forward(q, og, x0, x1, fk, plan)
                print(f"  one-plane weights of {','.join(names)}, {rounding}, {label}: increment {O.increment_rel_err(y, ref, x1).max().item():.2e}   "
                      f"per-channel {O.per_channel_rel_err(y, ref).max().item():.2e}", flush=True)
        return
    shipped = os.environ.get("GCNUM_PLAN", "edge_store,edge_hidden,static16")
    every = [] if os.environ.get("GCNUM_ONLY") else ["edge_store", "edge_hidden", "static16", "nodeterm16", "nodeterm_a", "node_a", "node_hidden", "embed_a", "out_a", "edge_w16", "edge_w1_16", "edge_w2_16"]
    extra = [(k, {k}) for k in os.environ.get("GCNUM_ONLY", "").split(",") if k]
    for name, plan in extra + [(k, {k}) for k in every] + [("shipped: " + shipped, set(shipped.split(","))), ("shipped + edge_w16", set(shipped.split(",")) | {"edge_w16"}), ("all", set(every))]:
        y = forward(p, og, x0, x1, fk, plan)
        print(f"  {name:60s} increment {O.increment_rel_err(y, ref, x1).max().item():.2e}   per-channel {O.per_channel_rel_err(y, ref).max().item():.2e}", flush=True)


if __name__ == "__main__":
    main()
