"""CPU restatement of the FourCastNet-v2-small (SFNO) forward step.  TEST INFRASTRUCTURE ONLY: imported by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg, never by the product path (skyrim_amd/).

PARITY UNPINNED.  The reference gets this network from ``earth2mip.networks.fcnv2_sm.load``
(/root/reference/skyrim/core/models/fourcastnet_v2.py:36-37; channel list :12-21; contract :24-28): a
``SphericalFourierNeuralOperatorNet`` (modulus / makani) over torch-harmonics' real spherical-harmonic transform,
normalised by ``global_means/stds``.  None of earth2mip, modulus, torch_harmonics or the checkpoint exists in this
environment and the reference's tests hold no numerical vectors for it, so this file restates the
PUBLIC definitions:

* torch-harmonics ``RealSHT`` / ``InverseRealSHT`` (norm="ortho", Condon-Shortley phase): x -> 2*pi * rfft(x, norm="forward")
  truncated to m < mmax, then the Legendre-Gauss / Clenshaw-Curtis quadrature against orthonormal associated Legendre
  functions; the inverse is the synthesis sum followed by irfft(norm="forward");
* the SFNO of Bonev et al. 2023 as implemented by makani's legacy ``sfnonet.py``: encoder MLP (1x1 convs), learned
  position embedding, ``num_layers`` blocks [instance norm -> spherical convolution (dhconv: one complex
  C_in x C_out matrix per degree l) -> + linear inner skip -> GELU -> instance norm -> MLP -> + residual], big skip
  (concat with the input), decoder MLP.  The first block analyses the 721x1440 equiangular grid and synthesises on an
  internal Legendre-Gauss grid (721 // scale_factor rows), the last block goes back out.

Hyper-parameters (``SfnoConfig`` defaults: embed 256, 8 layers, scale factor 3, MLP ratio 2) are the published
"sfno_73ch" small configuration as far as it can be recalled without the package; they are configuration, not code.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn.functional as F



class Shape:
    """Derived sizes of a configuration, computed HERE from its primary fields (n_lat, n_lon, scale_factor, num_layers, eps,
    out_chans) -- the oracle reads a configuration object as data and derives nothing from the product's code."""

    def __init__(self, cfg):
        self.n_lat, self.n_lon, self.num_layers, self.eps, self.out_chans = cfg.n_lat, cfg.n_lon, cfg.num_layers, cfg.eps, cfg.out_chans
        self.h, self.w = cfg.n_lat // cfg.scale_factor, cfg.n_lon // cfg.scale_factor      # internal Legendre-Gauss grid
        self.lmax = self.h                                                                 # triangular-ish truncation: l < h
        self.mmax = min(self.w // 2 + 1, self.lmax)                                        # m < min(w/2 + 1, lmax)


# ---- quadrature + Legendre functions (torch-harmonics quadrature.py / legendre.py) ---------------- #
def legendre_gauss(n: int):
    """colatitudes (ascending = north to south) and weights of the n-point Gauss-Legendre rule in cos(theta)."""
    x, w = np.polynomial.legendre.leggauss(n)
    return np.arccos(x[::-1]).copy(), w[::-1].copy()


def clenshaw_curtis(n: int):
    """n equiangular colatitudes INCLUDING both poles (the 721-row ERA5 grid) and Clenshaw-Curtis weights in cos(theta)."""
    theta = np.pi * np.arange(n) / (n - 1)
    w = np.zeros(n)
    nn = n - 1
    for j in range(n):
        s = 0.0
        for k in range(1, nn // 2 + 1):
            b = 1.0 if 2 * k == nn else 2.0
            s += b / (4.0 * k * k - 1.0) * math.cos(2.0 * k * theta[j])
        c = 1.0 if j in (0, nn) else 2.0
        w[j] = c / nn * (1.0 - s)
    return theta, w


def legendre_ortho(mmax: int, lmax: int, theta: np.ndarray) -> np.ndarray:
    """Orthonormal associated Legendre functions with Condon-Shortley phase: out[m, l, k] = Pbar_l^m(cos theta_k),
    zero for l < m;  2*pi * int Pbar_l^m Pbar_l'^m dcos(theta) = delta_ll'."""
    x, s = np.cos(theta), np.sin(theta)
    out = np.zeros((mmax, lmax, len(theta)))
    pmm = np.full_like(x, math.sqrt(1.0 / (4.0 * math.pi)))
    for m in range(mmax):
        if m > 0:
            pmm = -math.sqrt((2.0 * m + 1.0) / (2.0 * m)) * s * pmm
        if m < lmax:
            out[m, m] = pmm
        if m + 1 < lmax:
            out[m, m + 1] = math.sqrt(2.0 * m + 3.0) * x * pmm
        for l in range(m + 2, lmax):
            a = math.sqrt((4.0 * l * l - 1.0) / (l * l - m * m))
            b = math.sqrt(((l - 1.0) ** 2 - m * m) / (4.0 * (l - 1.0) ** 2 - 1.0))
            out[m, l] = a * (x * out[m, l - 1] - b * out[m, l - 2])
    return out


class SHT:
    """Real spherical-harmonic analysis / synthesis on an (n_lat, n_lon) grid, truncated to l < lmax, m < mmax."""

    def __init__(self, n_lat: int, n_lon: int, lmax: int, mmax: int, grid: str, dtype=torch.float64):
        theta, wq = legendre_gauss(n_lat) if grid == "legendre-gauss" else clenshaw_curtis(n_lat)
        p = legendre_ortho(mmax, lmax, theta)
        self.n_lat, self.n_lon, self.lmax, self.mmax = n_lat, n_lon, lmax, mmax
        self.analysis = torch.from_numpy(p * wq[None, None, :]).to(dtype)        # [m][l][lat]
        self.synthesis = torch.from_numpy(p).to(dtype)                           # [m][l][lat]

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x: (..., n_lat, n_lon) real -> (..., lmax, mmax) complex"""
        xf = 2.0 * math.pi * torch.fft.rfft(x.to(self.analysis.dtype), dim=-1, norm="forward")[..., : self.mmax]
        re = torch.einsum("...km,mlk->...lm", xf.real, self.analysis)
        im = torch.einsum("...km,mlk->...lm", xf.imag, self.analysis)
        return torch.complex(re, im)

    def inverse(self, c: torch.Tensor) -> torch.Tensor:
        """c: (..., lmax, mmax) complex -> (..., n_lat, n_lon) real"""
        re = torch.einsum("...lm,mlk->...km", c.real, self.synthesis)
        im = torch.einsum("...lm,mlk->...km", c.imag, self.synthesis)
        return torch.fft.irfft(torch.complex(re, im), n=self.n_lon, dim=-1, norm="forward")


# ---- the network ------------------------------------------------------------------------------------ #
def _conv1x1(x, w, b=None):
    """x: (C_in, H, W), w: (C_out, C_in)"""
    y = torch.einsum("oc,chw->ohw", w, x)
    return y if b is None else y + b[:, None, None]


def _instance_norm(x, g, b, eps):
    mu = x.mean(dim=(-2, -1), keepdim=True)
    var = ((x - mu) ** 2).mean(dim=(-2, -1), keepdim=True)
    return (x - mu) * torch.rsqrt(var + eps) * g[:, None, None] + b[:, None, None]


class Transforms:
    """The four SHTs of the network (sfnonet.py: trans_down, itrans_up, trans, itrans)."""

    def __init__(self, cfg, dtype=torch.float64):
        cfg = Shape(cfg)
        self.down = SHT(cfg.n_lat, cfg.n_lon, cfg.lmax, cfg.mmax, "equiangular", dtype)
        self.inner = SHT(cfg.h, cfg.w, cfg.lmax, cfg.mmax, "legendre-gauss", dtype)


def block(p: dict, prefix: str, x: torch.Tensor, fwd: SHT, inv: SHT, cfg, taps: dict | None = None):
    g = lambda n: p[prefix + n]  # noqa: E731
    xn = _instance_norm(x, g("norm0.weight"), g("norm0.bias"), cfg.eps)
    coef = fwd.forward(xn)                                                     # (C, L, M) complex
    residual = xn if (fwd.n_lat, fwd.n_lon) == (inv.n_lat, inv.n_lon) else inv.inverse(coef).to(x.dtype)
    w = torch.view_as_complex(g("filter.weight").to(coef.real.dtype).contiguous())    # (in, out, L)
    y = inv.inverse(torch.einsum("ilm,iol->olm", coef, w)).to(x.dtype)
    y = y + _conv1x1(residual, g("inner_skip.weight"), g("inner_skip.bias"))
    y = F.gelu(y)
    y = _instance_norm(y, g("norm1.weight"), g("norm1.bias"), cfg.eps)
    hdn = F.gelu(_conv1x1(y, g("mlp.fc1.weight"), g("mlp.fc1.bias")))
    y = _conv1x1(hdn, g("mlp.fc2.weight"), g("mlp.fc2.bias")) + residual
    if taps is not None:
        taps[prefix + "coef"] = coef
        taps[prefix + "out"] = y
    return y


def forward(params: dict, x: torch.Tensor, cfg, tr: Transforms | None = None, taps: dict | None = None) -> torch.Tensor:
    """One 6-h step: (in_chans, n_lat, n_lon) -> (out_chans, n_lat, n_lon), physical units in and out.  ``cfg``: any object with
    n_lat, n_lon, scale_factor, num_layers, eps, out_chans."""
    tr = tr or Transforms(cfg)
    cfg = Shape(cfg)
    p = params
    mean, std = p["norm.mean"][:, None, None], p["norm.std"][:, None, None]
    xin = (x - mean) / std
    y = F.gelu(_conv1x1(xin, p["encoder.fc1.weight"], p["encoder.fc1.bias"]))
    y = _conv1x1(y, p["encoder.fc2.weight"]) + p["pos_embed"]
    if taps is not None:
        taps["encoder"] = y
    for i in range(cfg.num_layers):
        fwd = tr.down if i == 0 else tr.inner
        inv = tr.down if i == cfg.num_layers - 1 else tr.inner
        y = block(p, f"blocks.{i}.", y, fwd, inv, cfg, taps)
    y = torch.cat([y, xin], dim=0)
    y = F.gelu(_conv1x1(y, p["decoder.fc1.weight"], p["decoder.fc1.bias"]))
    y = _conv1x1(y, p["decoder.fc2.weight"])
    return y * std[: cfg.out_chans] + mean[: cfg.out_chans]


def per_channel_rel_err(y: torch.Tensor, ref: torch.Tensor) -> torch.Tensor:
    y, ref = y.double(), ref.double()
    return (y - ref).abs().amax(dim=(-2, -1)) / ref.abs().amax(dim=(-2, -1)).clamp_min(1e-30)
