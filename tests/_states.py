"""States the built-in calibration state never looked like (test infrastructure): the default mode fits its one-plane weights to
`mean + sigma x 9x9-box-smoothed white noise` (skyrim_amd/pangu/engine.py: calibration_state); these have another spectrum, another
smoothing scale, or a meridional structure, in the same physical units."""
import math

import torch

from skyrim_amd.pangu.spec import PanguGeometry, channel_stats

KINDS = ("powerlaw", "smooth3", "smooth31", "meridional")


def _unit(n):
    n = n - n.flatten(1).mean(1)[:, None, None]
    return n / n.flatten(1).std(1)[:, None, None]


def _box(n, k):
    n = torch.nn.functional.pad(n[None], (k // 2, k // 2, 0, 0), mode="circular")
    n = torch.nn.functional.pad(n, (0, 0, k // 2, k // 2), mode="replicate")
    return torch.nn.functional.avg_pool2d(n, k, stride=1)[0]


def unit_field(g: PanguGeometry, kind: str, seed: int = 11) -> torch.Tensor:
    """(69, n_lat, n_lon) zero-mean unit-variance fields of the named kind."""
    gen = torch.Generator().manual_seed(seed)
    w = torch.randn(g.n_channels, g.n_lat, g.n_lon, generator=gen, dtype=torch.float64)
    if kind == "powerlaw":                       # isotropic k^-3 power spectrum (amplitude k^-1.5), periodic in both directions
        ky = torch.fft.fftfreq(g.n_lat, dtype=torch.float64)[:, None] * g.n_lat
        kx = torch.fft.rfftfreq(g.n_lon, dtype=torch.float64)[None, :] * g.n_lon
        k = torch.sqrt(ky * ky + kx * kx)
        amp = torch.where(k > 0, k.clamp(min=1.0) ** -1.5, torch.zeros_like(k))
        n = torch.fft.irfft2(torch.fft.rfft2(w) * amp, s=(g.n_lat, g.n_lon))
    elif kind == "smooth3":
        n = _box(w, 3)
    elif kind == "smooth31":
        n = _box(w, 31)
    elif kind == "meridional":                   # a zonal-mean profile (equator-to-pole gradient + a jet-like bump) + a third of smooth noise
        lat = torch.linspace(90.0, -90.0, g.n_lat, dtype=torch.float64) * math.pi / 180.0
        c = torch.arange(g.n_channels, dtype=torch.float64)[:, None]
        prof = torch.cos(2.0 * lat)[None, :] * (1.0 + 0.05 * c) + 0.6 * torch.exp(-((lat[None, :] - 0.7 + 0.01 * c) / 0.15) ** 2)
        n = _unit(prof[:, :, None].expand(-1, -1, g.n_lon).contiguous() + 0.0 * w) + 0.33 * _unit(_box(w, 9))
    else:
        raise ValueError(kind)
    return _unit(n).float()


def state(g: PanguGeometry, kind: str, seed: int = 11) -> torch.Tensor:
    mean, std = channel_stats()
    return (mean[:, None, None] + std[:, None, None] * unit_field(g, kind, seed)).contiguous()


def gain_one_params(params: dict, embed_scale: float = 3.5, recover_scale: float = 3.5) -> dict:
    """The synthetic parameter set with its patch embedding and patch recovery weights scaled so that the 6-h map neither damps nor
    amplifies a small perturbation of the state (random-init weights damp it ~12x per step: measured gain 0.086; the gain is linear in
    both scales: x3.5 each -> ~0.95).  Everything between the two (LayerNorm-bounded residual blocks) is unchanged."""
    out = dict(params)
    for k, v in params.items():
        if k.endswith("weight") and k.startswith("embed.conv"):
            out[k] = v * embed_scale
        if k.endswith("weight") and k.startswith("recover.conv"):
            out[k] = v * recover_scale
    return out
