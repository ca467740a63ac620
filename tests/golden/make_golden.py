"""Generates tests/golden/pangu_toy_49x192.npz with the CPU oracle (oracle/pangu_oracle.py).

SELF-ORACLE, REFERENCE PARITY UNPINNED: the reference's arithmetic for this path lives in
earth2mip/onnxruntime + downloaded ONNX weights, none of which exist in the build container, and the
reference's own tests hold no numerical vector for it.  These vectors pin the oracle
against silent drift; inputs are regenerated from seeds, outputs are stored.

    python tests/golden/make_golden.py
"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import pangu_oracle as O  # noqa: E402
from skyrim_amd.pangu.spec import PanguGeometry, init_synthetic, synthetic_state  # noqa: E402

if __name__ == "__main__":
    torch.set_num_threads(8)
    g = PanguGeometry(49, 192)
    params = init_synthetic(g, 0)
    x = synthetic_state(g, 0)
    taps = {}
    y = O.forward(params, x.double(), taps=taps)
    y2 = O.forward(params, y)
    out = {
        "state_in_sub": x[:, ::6, ::16].numpy(),
        "step1_sub": y[:, ::6, ::16].float().numpy(),                 # (69, 9, 12)
        "step2_sub": y2[:, ::6, ::16].float().numpy(),
        "step1_channel_mean": y.flatten(1).mean(1).numpy(),
        "step1_channel_absmax": y.flatten(1).abs().max(1).values.numpy(),
        "embed_sub": taps["embed"][::97, ::7].float().numpy(),
        "layer1_block1_sub": taps["layer1.block1"][::97, ::7].float().numpy(),
        "down_sub": taps["down"][::31, ::11].float().numpy(),
        "layer3_sub": taps["layer3"][::31, ::11].float().numpy(),
        "up_sub": taps["up"][::97, ::7].float().numpy(),
        "layer4_sub": taps["layer4"][::97, ::7].float().numpy(),
        "position_index_sub": O.position_index()[::5, ::7].numpy(),
    }
    np.savez_compressed(Path(__file__).parent / "pangu_toy_49x192.npz", **out)
    print({k: v.shape for k, v in out.items()})
