"""Shapes, parameter layout and synthetic inputs of the SFNO (FourCastNet v2-small) step.

The network is modulus / makani's legacy ``SphericalFourierNeuralOperatorNet`` as loaded by
``earth2mip.networks.fcnv2_sm.load`` (/root/reference/skyrim/core/models/fourcastnet_v2.py:36-37); the default
hyper-parameters are the published small 73-channel configuration as far as it can be recalled without the package
(neither the package nor the checkpoint is available here) -- they are configuration, not code.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F

# channel order of the reference wrapper (fourcastnet_v2.py:12-21)
_LEVELS = [50, 100, 150, 200, 250, 300, 400, 500, 600, 700, 850, 925, 1000]
CHANNELS = ["u10m", "v10m", "u100m", "v100m", "t2m", "sp", "msl", "tcwv"] + [f"{v}{l}" for v in "uvztr" for l in _LEVELS]


@dataclass(frozen=True)
class SfnoConfig:
    n_lat: int = 721
    n_lon: int = 1440
    in_chans: int = 73
    out_chans: int = 73
    embed_dim: int = 256
    num_layers: int = 8
    scale_factor: int = 3
    mlp_ratio: int = 2
    eps: float = 1e-6            # instance norm

    @property
    def h(self):                 # internal (Legendre-Gauss) grid
        return self.n_lat // self.scale_factor

    @property
    def w(self):
        return self.n_lon // self.scale_factor

    @property
    def lmax(self):              # degrees l < lmax and orders m < mmax are kept
        return self.h

    @property
    def mmax(self):
        return min(self.w // 2 + 1, self.lmax)


def param_spec(cfg: SfnoConfig) -> list[tuple[str, tuple]]:
    e, hid = cfg.embed_dim, cfg.embed_dim * cfg.mlp_ratio
    spec = [("norm.mean", (cfg.in_chans,)), ("norm.std", (cfg.in_chans,)),
            ("encoder.fc1.weight", (e, cfg.in_chans)), ("encoder.fc1.bias", (e,)), ("encoder.fc2.weight", (e, e)),
            ("pos_embed", (e, cfg.n_lat, cfg.n_lon))]
    for i in range(cfg.num_layers):
        p = f"blocks.{i}."
        spec += [(p + "norm0.weight", (e,)), (p + "norm0.bias", (e,)),
                 (p + "filter.weight", (e, e, cfg.lmax, 2)),           # [in][out][l][re, im]  (dhconv)
                 (p + "inner_skip.weight", (e, e)), (p + "inner_skip.bias", (e,)),
                 (p + "norm1.weight", (e,)), (p + "norm1.bias", (e,)),
                 (p + "mlp.fc1.weight", (hid, e)), (p + "mlp.fc1.bias", (hid,)),
                 (p + "mlp.fc2.weight", (e, hid)), (p + "mlp.fc2.bias", (e,))]
    spec += [("decoder.fc1.weight", (e, e + cfg.in_chans)), ("decoder.fc1.bias", (e,)), ("decoder.fc2.weight", (cfg.out_chans, e))]
    return spec


def channel_stats(cfg: SfnoConfig):
    return torch.linspace(-5.0, 300.0, cfg.in_chans), torch.linspace(1.0, 30.0, cfg.in_chans)


def init_synthetic(cfg: SfnoConfig, seed: int = 0) -> dict:
    """Seeded random parameters with trained-network magnitudes (no checkpoint in this environment)."""
    gen = torch.Generator().manual_seed(seed)
    mean, std = channel_stats(cfg)
    out = {}
    for name, shape in param_spec(cfg):
        if name == "norm.mean":
            t = mean
        elif name == "norm.std":
            t = std
        elif name.endswith("norm0.weight") or name.endswith("norm1.weight"):
            t = 1.0 + 0.05 * torch.randn(shape, generator=gen)
        elif name.endswith(".bias") or name == "pos_embed":
            t = 0.02 * torch.randn(shape, generator=gen)
        elif name.endswith("filter.weight"):
            t = torch.randn(shape, generator=gen) * math.sqrt(1.0 / shape[0])
        else:
            t = torch.randn(shape, generator=gen) * math.sqrt(1.0 / shape[-1])
        out[name] = t.float().contiguous()
    return out


def synthetic_state(cfg: SfnoConfig, seed: int = 0) -> torch.Tensor:
    """(in_chans, n_lat, n_lon) fp32 state: per-channel mean + std * smooth noise (9x9 box filter, periodic in lon)."""
    gen = torch.Generator().manual_seed(1000 + seed)
    z = torch.randn(cfg.in_chans, cfg.n_lat, cfg.n_lon, generator=gen)
    z = F.avg_pool2d(F.pad(z[None], (4, 4, 0, 0), mode="circular"), (1, 9), stride=1)[0]
    z = F.avg_pool2d(F.pad(z[None], (0, 0, 4, 4), mode="replicate"), (9, 1), stride=1)[0] * 9.0
    mean, std = channel_stats(cfg)
    return (mean[:, None, None] + std[:, None, None] * z).float().contiguous()


def flops_per_step(cfg: SfnoConfig) -> float:
    """Algorithmic FLOPs of one step: FFTs as 5 N log2 N, Legendre / dhconv sums over the (l >= m) pairs only."""
    e, hid, hw_out = cfg.embed_dim, cfg.embed_dim * cfg.mlp_ratio, cfg.n_lat * cfg.n_lon
    lm = sum(cfg.lmax - m for m in range(cfg.mmax))
    total = 2.0 * hw_out * (cfg.in_chans * e + e * e) + 2.0 * hw_out * ((e + cfg.in_chans) * e + e * cfg.out_chans)
    for i in range(cfg.num_layers):
        n_in = (cfg.n_lat, cfg.n_lon) if i == 0 else (cfg.h, cfg.w)
        n_out = (cfg.n_lat, cfg.n_lon) if i == cfg.num_layers - 1 else (cfg.h, cfg.w)
        fft = lambda n: 2.5 * n[1] * math.log2(n[1]) * n[0] * e  # noqa: E731
        leg = lambda n: 4.0 * lm * n[0] * e                      # noqa: E731
        n_inv = 2 if n_in != n_out else 1                        # the resampled residual is a second synthesis
        total += fft(n_in) + leg(n_in) + 8.0 * lm * e * e + n_inv * (leg(n_out) + fft(n_out))
        total += 2.0 * n_out[0] * n_out[1] * (e * e + 2 * e * hid)
    return total
