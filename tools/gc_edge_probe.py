"""GPU box: time the fused GraphCast kernels on synthetic full-size inputs and print where a tile's clocks go (skgc_edge_desc::probe).

    SKYRIM_GRAPHCAST_LIB=skyrim_amd/lib/variants/libgc_<v>.so python tools/gc_edge_probe.py [tiles_fc1 tiles_static node_rows]

Probe variants are built by tools/build_gc_variants.sh (-DSKP_PROBES=<bits>).  Measurement only; results are not checked here.
"""
import os
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from skyrim_amd import ops  # noqa: E402
from skyrim_amd.graphcast import fused as fz  # noqa: E402

L = 512


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    return sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(n))[n // 2]


def phases(probe):
    p = probe.cpu().numpy().astype(np.float64)
    d = np.diff(p, axis=1)[:, :5]
    names = ["prologue", "phase1", "phase2", "ln+scan", "exchange+store"]
    med = np.median(d, axis=0)
    return " ".join(f"{n}={m:.0f}" for n, m in zip(names, med)) + f"  total={np.median(p[:, 5] - p[:, 0]):.0f} ticks"


def main():
    a = [int(v) for v in sys.argv[1:4]]
    t_fc1, t_static, node_rows = a + [2640, 12288, 40962][len(a):]
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(0)
    print("lib:", os.environ.get("SKYRIM_GRAPHCAST_LIB", "default"))
    n_nodes = int(os.environ.get("GCP_NODES", "40962"))
    print("nodes", n_nodes)
    w2f = fz.prep_w2_fragments((torch.randn(L, L, generator=gen) / L ** 0.5).to(dev))
    planes = int(os.environ.get("GCP_W1_PLANES", "1"))
    w1f = fz.prep_w1_fragments((torch.randn(L, L, generator=gen) / L ** 0.5).to(dev), planes)
    tab = [torch.randn(L, generator=gen).to(dev) * 0.1 for _ in range(3)]
    terms = torch.randn(n_nodes, 2 * L, generator=gen).to(dev)
    for name, tiles, fc1 in (("fc1 (processor edges)", t_fc1, True), ("static, 2 terms (mesh->grid)", t_static, False)):
        R = tiles * 128
        recv = np.repeat(np.arange(R // 8 + 1), 8)[:R] % n_nodes
        recv = np.sort(recv).astype(np.int32)
        send = torch.randint(0, n_nodes, (R,), generator=gen).int().to(dev)
        recv_d = torch.from_numpy(recv).to(dev)
        e = (torch.randn(R // 16, 16 * L, generator=gen) * 0.5).half().reshape(-1).to(dev)
        agg = torch.zeros(n_nodes, L, device=dev)
        heads = torch.zeros(tiles, L, device=dev)
        probe = torch.zeros(tiles, 8, dtype=torch.int64, device=dev)
        args = (e, e if fc1 else None, [terms, terms], [0, L], [2 * L, 2 * L], [send, recv_d], recv_d, w1f if fc1 else None, w2f, *tab, agg, heads, R)
        ms = timed(lambda: ops.hip.gc_edge_update(*args, None, planes))
        ops.hip.gc_edge_update(*args, probe, planes)
        torch.cuda.synchronize()
        mf = 2.0 * R * L * L * ((2 + planes) if fc1 else 2)
        print(f"{name}: {tiles} tiles, {ms:.3f} ms, {mf / ms / 1e9:.0f} TFLOP/s executed (W1 planes {planes}); {ms * 1e3 / (tiles / 256):.1f} us per tile round;  {phases(probe)}")
    for ns in (1, 2):
        srcs = [torch.randn(node_rows, L, generator=gen).to(dev) for _ in range(ns)]
        n1 = fz.prep_w1_node((torch.randn(L, L * ns, generator=gen) / (L * ns) ** 0.5).to(dev))
        out = torch.empty(node_rows, L, device=dev)
        b1 = torch.zeros(L, device=dev)
        ms = timed(lambda: ops.hip.gc_node_mlp(srcs, [0] * ns, [L] * ns, n1, w2f, b1, *tab, srcs[0], 0, L, out, 0, L, node_rows))
        mf = 2.0 * node_rows * L * L * (ns + 1) * 3
        print(f"node mlp, {ns} source(s): {node_rows} rows, {ms:.3f} ms, {mf / ms / 1e9:.0f} TFLOP/s executed (3 terms)")


if __name__ == "__main__":
    main()
