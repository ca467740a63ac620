"""The fused GraphCast kernels (include/skyrim_graphcast.h ABI v4: skgc_edge_update, skgc_segment_fixup, skgc_node_mlp) against float64
restatements of the same arithmetic on the CPU (torch), through the custom-op boundary.  The restatement rounds where the kernels round:
edge operands and hidden activations of the EDGE kernel to one fp16 plane; the node kernel keeps hi/lo pairs (fp32-class)."""
import numpy as np
import pytest
import torch

from skyrim_amd.graphcast import fused as fz

pytestmark = pytest.mark.gpu
L = 512


def f16(x):
    return x.to(torch.float16).to(torch.float64)


def _ln(z, gamma, beta):
    return torch.nn.functional.layer_norm(z, (L,), gamma, beta, 1e-5)


def _edge_case(seed, lens, n_nodes):
    gen = torch.Generator().manual_seed(seed)
    recv_e = np.repeat(np.arange(len(lens)) * 2 + 1, lens)
    E = len(recv_e)
    send_e = torch.randint(0, n_nodes, (E,), generator=gen).numpy()
    r = lambda *s: torch.randn(*s, generator=gen, dtype=torch.float64)  # noqa: E731
    d = dict(recv_e=recv_e, send_e=send_e, e=r(E, L), w1e=r(L, L) / L ** 0.5, w2=r(L, L) / L ** 0.5, b2=0.1 * r(L), gamma=1 + 0.1 * r(L), beta=0.1 * r(L),
             ts=r(n_nodes, L), tr=r(n_nodes, L))
    row_edge = fz.pack_segments(recv_e)
    ok = row_edge >= 0
    re = np.where(ok, row_edge, 0)
    d.update(row_edge=row_edge, ok=ok, re=re, recv=np.where(ok, recv_e[re], -1).astype(np.int32), send=np.where(ok, send_e[re], -1).astype(np.int32))
    return d


@pytest.mark.parametrize("has_fc1,planes", [(True, 2), (True, 1), (False, 2)])
def test_edge_update_and_receiver_sum_vs_float64(has_fc1, planes):
    from skyrim_amd import ops
    rng = np.random.default_rng(5)
    lens = np.r_[rng.integers(1, 43, size=60), 300, rng.integers(1, 43, size=30), 1, 1, 130]
    n_nodes = 2 * len(lens) + 3
    c = _edge_case(11 if has_fc1 else 12, lens, n_nodes)
    dev = torch.device("cuda:0")
    R = len(c["row_edge"])
    upos = torch.from_numpy(fz.unit_at_pos())
    t2 = torch.cat([c["ts"][:, upos], c["tr"][:, upos]], dim=1).float().contiguous().to(dev)       # [nodes][1024], "pos" order
    packed = torch.where(torch.from_numpy(c["ok"])[:, None], c["e"][c["re"]], torch.zeros(1, dtype=torch.float64))
    recv_d, send_d = torch.from_numpy(c["recv"]).to(dev), torch.from_numpy(c["send"]).to(dev)
    w2f = fz.prep_w2_fragments(c["w2"].float().to(dev))
    b2, gamma, beta = (c[k].float().to(dev) for k in ("b2", "gamma", "beta"))
    agg = torch.full((n_nodes, L), float("nan"), device=dev)
    heads = torch.zeros(R // 128, L, device=dev)
    recv_e, send_e = torch.from_numpy(c["recv_e"]), torch.from_numpy(c["send_e"])
    if has_fc1:
        e_b = fz.to_blocked_f16(packed.float().to(dev))
        e_out = e_b.clone()
        w1f = fz.prep_w1_fragments(c["w1e"].float().to(dev), planes)
        ops.hip.gc_edge_update(e_b, e_out, [t2, t2], [0, L], [2 * L, 2 * L], [send_d, recv_d], recv_d, w1f, w2f, b2, gamma, beta, agg, heads, R, None, planes)
        x = f16(c["e"])
        w1 = c["w1e"].float().double() if planes == 2 else f16(c["w1e"].float())                    # one plane: W_e itself is rounded to fp16
        pre = x @ w1.T + c["ts"].float().double()[send_e] + c["tr"].float().double()[recv_e]
    else:
        e_b = fz.to_blocked_f16(packed[:, upos].float().to(dev))                                   # the prepared term, "pos" columns
        e_out = None
        ops.hip.gc_edge_update(e_b, None, [t2, t2], [0, L], [2 * L, 2 * L], [send_d, recv_d], recv_d, None, w2f, b2, gamma, beta, agg, heads, R)
        pre = f16(c["e"]) + c["ts"].float().double()[send_e] + c["tr"].float().double()[recv_e]
    nodes, first, tiles = fz.continuation_list(c["recv"])
    assert len(nodes) == 2                                                                         # the 300-run and the 130-run continue over tiles
    i32 = lambda a: torch.from_numpy(a).to(dev)  # noqa: E731
    ops.hip.gc_segment_fixup(agg, heads, i32(nodes), i32(first), i32(tiles))
    torch.cuda.synchronize()
    h = f16(torch.nn.functional.silu(pre))
    y = _ln(h @ c["w2"].float().double().T + c["b2"].float().double(), c["gamma"].float().double(), c["beta"].float().double())
    ref = torch.zeros(n_nodes, L, dtype=torch.float64).index_add_(0, recv_e, y)
    got = agg.cpu().double()
    touched = torch.unique(recv_e)
    rest = torch.ones(n_nodes, dtype=torch.bool); rest[touched] = False
    assert torch.isnan(got[rest]).all()                                                            # receivers without rows are never written
    assert ((got[touched] - ref[touched]).abs().max() / ref.abs().max()).item() < 3e-4
    if has_fc1:
        out = fz.from_blocked_f16(e_out.cpu(), R, L)[torch.from_numpy(c["ok"])].double()
        want = f16(x + y)
        assert ((out - want).abs().max() / want.abs().max()).item() < 2e-3                          # one fp16 ulp where a rounding boundary is crossed
        assert (out - want).abs().mean().item() < 1e-4
        pad = fz.from_blocked_f16(e_out.cpu(), R, L)[~torch.from_numpy(c["ok"])]
        assert float(pad.abs().max()) == 0.0                                                       # padding rows stay zero
    # deterministic: a second call gives the same bits
    agg2 = torch.full_like(agg, float("nan"))
    heads2 = torch.zeros_like(heads)
    ops.hip.gc_edge_update(e_b, e_out.clone() if has_fc1 else None, [t2, t2], [0, L], [2 * L, 2 * L], [send_d, recv_d], recv_d, w1f if has_fc1 else None, w2f,
                           b2, gamma, beta, agg2, heads2, R, None, planes)
    ops.hip.gc_segment_fixup(agg2, heads2, i32(nodes), i32(first), i32(tiles))
    assert torch.equal(torch.nan_to_num(agg2), torch.nan_to_num(agg))


@pytest.mark.parametrize("n_src,rows", [(1, 200), (2, 333), (2, 64)])
def test_node_mlp_vs_float64(n_src, rows):
    from skyrim_amd import ops
    gen = torch.Generator().manual_seed(20 + n_src)
    r = lambda *s: torch.randn(*s, generator=gen, dtype=torch.float64)  # noqa: E731
    dev = torch.device("cuda:0")
    srcs = [3.0 * r(rows, L).float() for _ in range(n_src)]
    w1, w2 = (r(L, L * n_src) / (L * n_src) ** 0.5).float(), (r(L, L) / L ** 0.5).float()
    b1, b2, gamma, beta = (0.1 * r(L)).float(), (0.1 * r(L)).float(), (1 + 0.1 * r(L)).float(), (0.1 * r(L)).float()
    x = torch.cat(srcs, dim=1).double()
    y = _ln(torch.nn.functional.silu(x @ w1.double().T + b1.double()) @ w2.double().T + b2.double(), gamma.double(), beta.double())
    want = srcs[0].double() + y
    sd = [s.to(dev) for s in srcs]
    w1f, w2f = fz.prep_w1_node(w1.to(dev)), fz.prep_w2_fragments(w2.to(dev))
    tab = [t.to(dev) for t in (b1, b2, gamma, beta)]
    out = torch.full((rows + 3, L), 7.0, device=dev)                                               # rows beyond `rows` must stay untouched
    ops.hip.gc_node_mlp(sd, [0] * n_src, [L] * n_src, w1f, w2f, *tab, sd[0], 0, L, out, 0, L, rows)
    torch.cuda.synchronize()
    assert float((out[rows:] - 7.0).abs().max()) == 0.0
    assert ((out[:rows].cpu().double() - want).abs().max() / want.abs().max()).item() < 3e-6
    # in place on the residual source, no residual
    ops.hip.gc_node_mlp(sd, [0] * n_src, [L] * n_src, w1f, w2f, *tab, sd[0], 0, L, sd[0], 0, L, rows)
    torch.cuda.synchronize()
    assert ((sd[0].cpu().double() - want).abs().max() / want.abs().max()).item() < 3e-6


def test_hi_lo_split_is_consistent_for_every_element():
    """Identity weights: the kernel's output is LayerNorm(swish(x)) through the hi/lo planes of x and of the hidden activation.  With a split
    whose lo plane is not derived from the STORED hi plane (hipcc converts twice, with v_cvt_pk_f16_f32 and v_cvt_f16_f32, and the two disagree
    for values within an fp32 ulp of an fp16 grid point) single elements come out one fp16 ulp off -- 17 of 2 M here before csrc/common.h:split8
    made the hi plane opaque."""
    from skyrim_amd import ops
    gen = torch.Generator().manual_seed(21)
    rows, dev = 4096, torch.device("cuda:0")
    x = (3.0 * torch.randn(rows, L, generator=gen, dtype=torch.float64)).float()
    eye, zero, one = torch.eye(L), torch.zeros(L), torch.ones(L)
    w1f, w2f = fz.prep_w1_node(eye.to(dev)), fz.prep_w2_fragments(eye.to(dev))
    out = torch.zeros(rows, L, device=dev)
    ops.hip.gc_node_mlp([x.to(dev)], [0], [L], w1f, w2f, zero.to(dev), zero.to(dev), one.to(dev), zero.to(dev), None, 0, L, out, 0, L, rows)
    torch.cuda.synchronize()
    h = torch.nn.functional.silu(x.double())
    mean, std = h.mean(1, keepdim=True), (h.var(1, unbiased=False, keepdim=True) + 1e-5).sqrt()
    d = (out.cpu().double() * std + mean - h).abs()                       # LayerNorm undone with the exact statistics: per-element view
    assert int((d > 2e-4).sum()) == 0 and d.max().item() < 3e-5


def test_fused_engine_matches_the_round3_kernel_sequence(monkeypatch):
    """latent 512 on a small grid: the fused path (default) against SKGC_UNFUSED=1 (fp32 latents, three MFMA terms) and against the oracle."""
    from oracle import graphcast_graph as OG
    from oracle import graphcast_oracle as O
    from skyrim_amd.graphcast.engine import GraphcastEngine
    from skyrim_amd.graphcast.spec import GraphcastConfig, forcings, init_synthetic, synthetic_states
    cfg = GraphcastConfig(n_lat=35, n_lon=72, splits=3, latent=512, steps=3)
    p = init_synthetic(cfg, 0)
    x0, x1 = synthetic_states(cfg, 0)
    fk = forcings(cfg, 1000.0)
    eng = GraphcastEngine(cfg, "cuda:0")
    assert eng.fused
    eng.load_params(p)
    a = eng.step(x0.cuda(), x1.cuda(), fk.cuda()).cpu()
    a2 = eng.step(x0.cuda(), x1.cuda(), fk.cuda()).cpu()
    assert torch.equal(a, a2)                                                                       # deterministic receiver sums
    monkeypatch.setenv("SKGC_UNFUSED", "1")
    old = GraphcastEngine(cfg, "cuda:0")
    assert not old.fused
    old.load_params(p)
    b = old.step(x0.cuda(), x1.cuda(), fk.cuda()).cpu()
    ref = O.forward(p, OG.build(cfg.n_lat, cfg.n_lon, cfg.splits), x0, x1, fk)
    e_new, e_old = O.increment_rel_err(a, ref, x1).max().item(), O.increment_rel_err(b, ref, x1).max().item()
    assert e_old < 1e-4 and e_new < 4e-4, (e_new, e_old)
    assert O.per_channel_rel_err(a, ref).max().item() < 1e-5
