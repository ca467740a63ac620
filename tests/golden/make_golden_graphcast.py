"""Generates tests/golden/graphcast_tiny_33x64.npz with the CPU oracle (oracle/graphcast_oracle.py).

SELF-ORACLE, REFERENCE PARITY UNPINNED: the reference's GraphCast arithmetic lives in earth2mip / DeepMind's JAX package and
a downloaded checkpoint, none of which exist in the build container, and the reference's tests hold no numerical vector for
it.  These vectors pin the oracle (and the graph construction) against silent drift.

    python tests/golden/make_golden_graphcast.py
"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import graphcast_oracle as O  # noqa: E402
from oracle import graphcast_graph as OG  # noqa: E402
from skyrim_amd.graphcast.spec import GraphcastConfig, forcings, init_synthetic, synthetic_states  # noqa: E402

TINY = dict(n_lat=33, n_lon=64, splits=2, latent=32, steps=3)

if __name__ == "__main__":
    torch.set_num_threads(8)
    cfg = GraphcastConfig(**TINY)
    g = OG.build(cfg.n_lat, cfg.n_lon, cfg.splits)          # the oracle's own graph construction
    p = init_synthetic(cfg, 0)
    x0, x1 = synthetic_states(cfg, 0)
    f = forcings(cfg, 1000.0)
    taps = {}
    y = O.forward(p, g, x0, x1, f, taps=taps)
    np.savez_compressed(Path(__file__).with_name("graphcast_tiny_33x64.npz"),
                        x_prev_sub=x0[:, ::4, ::8].numpy(), x_cur_sub=x1[:, ::2, ::4].numpy(), forcing_sub=f[:, ::4, ::8].numpy(), step1_sub=y[:, ::2, ::4].numpy(),
                        increment_absmax=(y - x1).abs().amax(dim=(1, 2)).numpy(),
                        encoder_vm_sub=taps["encoder.vm"][::7].numpy(), processor_vm_sub=taps["processor.vm"][::7].numpy(),
                        mesh_edges=g.mesh_edges.astype(np.int32), g2m_edges_sub=g.g2m_edges[::13].astype(np.int32),
                        m2g_senders_sub=g.m2g_edges[::11, 0].astype(np.int32), mesh_edge_feat_sub=g.mesh_edge_feat[::9])
    print("written", Path(__file__).with_name("graphcast_tiny_33x64.npz"))
