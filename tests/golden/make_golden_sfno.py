"""Generates tests/golden/sfno_tiny_33x64.npz with the CPU oracle (oracle/sfno_oracle.py).

SELF-ORACLE, REFERENCE PARITY UNPINNED: the reference's SFNO arithmetic lives in earth2mip / modulus / torch-harmonics and
a downloaded checkpoint, none of which exist in the build container, and the reference's tests hold no numerical vector
for it.  These vectors pin the oracle against silent drift; inputs are regenerated from seeds.

    python tests/golden/make_golden_sfno.py
"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import sfno_oracle as O  # noqa: E402
from skyrim_amd.sfno.spec import SfnoConfig, init_synthetic, synthetic_state  # noqa: E402

TINY = dict(n_lat=33, n_lon=64, in_chans=5, out_chans=5, embed_dim=16, num_layers=3, scale_factor=2)

if __name__ == "__main__":
    torch.set_num_threads(8)
    cfg = SfnoConfig(**TINY)
    params, x = init_synthetic(cfg, 0), synthetic_state(cfg, 0)
    taps = {}
    y = O.forward(params, x, cfg, taps=taps)
    y2 = O.forward(params, y, cfg)
    sht = O.SHT(cfg.n_lat, cfg.n_lon, cfg.lmax, cfg.mmax, "equiangular")
    coef = sht.forward(x[:2].double())
    np.savez_compressed(Path(__file__).with_name("sfno_tiny_33x64.npz"),
                        state_in=x.numpy(), step1=y.numpy(), step2=y2.numpy(),
                        encoder_sub=taps["encoder"][::3, ::4, ::8].numpy(), block0_out_sub=taps["blocks.0.out"][::3, ::2, ::4].numpy(),
                        sht_coef_re=coef.real.numpy().astype(np.float32), sht_coef_im=coef.imag.numpy().astype(np.float32))
    print("written", Path(__file__).with_name("sfno_tiny_33x64.npz"))
