"""Pangu-Weather 6-h network: geometry, parameter inventory and synthetic initialisation.

Host-side description of the model that the HIP engine (csrc/) executes.  The
reference obtains this network as an opaque ONNX graph through
``earth2mip.networks.pangu.load`` (/root/reference/skyrim/core/models/pangu.py:45-46);
here the architecture constants follow the public Pangu-Weather pseudocode and the
parameter names mirror a PyTorch state dict so that real weights can be mapped in later.

Nothing in this file touches the CPU oracle (oracle/ is test infrastructure).
"""
from __future__ import annotations

from dataclasses import dataclass
from functools import cached_property

import torch

# channel order of the reference wrapper, skyrim/core/models/pangu.py:6-13
LEVELS = [1000, 925, 850, 700, 600, 500, 400, 300, 250, 200, 150, 100, 50]
UPPER_VARS = ["z", "q", "t", "u", "v"]
SURFACE_VARS = ["msl", "u10m", "v10m", "t2m"]
CHANNELS = [f"{v}{l}" for v in UPPER_VARS for l in LEVELS] + SURFACE_VARS

WINDOW = (2, 6, 12)
WIN_TOKENS = WINDOW[0] * WINDOW[1] * WINDOW[2]          # 144
PATCH = (2, 4, 4)
DIM = 192
HEADS = (6, 12, 12, 6)
DEPTHS = (2, 6, 6, 2)
HEAD_DIM = 32
BIAS_TABLE_ROWS = (2 * WINDOW[2] - 1) * WINDOW[1] ** 2 * WINDOW[0] ** 2   # 3312
N_CONST_MASKS = 3


def _pad_to(n: int, mult: int, pad: str = "centre") -> tuple[int, int]:
    """(padded, front): "centre" puts total // 2 zeros in front, "back" none (all padding behind the data)."""
    padded = (n + mult - 1) // mult * mult
    return padded, ((padded - n) // 2 if pad == "centre" else 0)


@dataclass(frozen=True)
class PanguGeometry:
    n_lat: int = 721
    n_lon: int = 1440
    pad: str = "centre"          # zero-padding placement, one of the conventions the pseudocode leaves open ("centre" | "back")

    def __post_init__(self):
        if self.pad not in ("centre", "back"):
            raise ValueError("pad is 'centre' or 'back'")
        if self.n_lon % (PATCH[2] * 2 * WINDOW[2]) != 0:
            raise ValueError("n_lon must be a multiple of 96 (patch 4 x merge 2 x window 12)")
        if self.n_lat < 8:
            raise ValueError("n_lat too small")

    # ---- input grid ------------------------------------------------------ #
    @cached_property
    def lat_padded(self): return _pad_to(self.n_lat, PATCH[1], self.pad)[0]
    @cached_property
    def lat_pad_top(self): return _pad_to(self.n_lat, PATCH[1], self.pad)[1]
    @property
    def n_levels(self): return len(LEVELS)
    @property
    def n_channels(self): return len(CHANNELS)
    # ---- token grids ----------------------------------------------------- #
    @property
    def Z(self): return 1 + (self.n_levels + 1) // 2        # 8 (surface slab + 7)
    @cached_property
    def H1(self): return self.lat_padded // PATCH[1]
    @cached_property
    def W1(self): return self.n_lon // PATCH[2]
    @cached_property
    def H2(self): return _pad_to(self.H1, 2, self.pad)[0] // 2
    @cached_property
    def W2(self): return self.W1 // 2

    def res(self, layer: int): return (self.Z, self.H1, self.W1) if layer in (1, 4) else (self.Z, self.H2, self.W2)
    def dim(self, layer: int): return DIM if layer in (1, 4) else 2 * DIM
    def tokens(self, layer: int):
        z, h, w = self.res(layer); return z * h * w
    def padded_lat(self, layer: int): return _pad_to(self.res(layer)[1], WINDOW[1], self.pad)[0]
    def pad_top(self, layer: int): return _pad_to(self.res(layer)[1], WINDOW[1], self.pad)[1]
    def n_windows(self, layer: int):
        z, _, w = self.res(layer)
        return (z // WINDOW[0]) * (self.padded_lat(layer) // WINDOW[1]) * (w // WINDOW[2])
    def window_types(self, layer: int):
        return (self.Z // WINDOW[0]) * (self.padded_lat(layer) // WINDOW[1])

    @property
    def lat(self): return [90.0 - 180.0 * i / (self.n_lat - 1) for i in range(self.n_lat)]
    @property
    def lon(self): return [360.0 * j / self.n_lon for j in range(self.n_lon)]


def param_spec(g: PanguGeometry) -> list[tuple[str, tuple[int, ...]]]:
    """Ordered (name, shape) list: the canonical fp32 'master' parameter blob layout
    handed to skpangu_prepare() (include/skyrim_pangu.h)."""
    C = DIM
    spec: list[tuple[str, tuple[int, ...]]] = [
        ("norm.mean", (g.n_channels,)),
        ("norm.std", (g.n_channels,)),
        ("const_masks", (N_CONST_MASKS, g.n_lat, g.n_lon)),
        ("embed.conv.weight", (C, len(UPPER_VARS), *PATCH)),
        ("embed.conv.bias", (C,)),
        ("embed.conv_surface.weight", (C, len(SURFACE_VARS) + N_CONST_MASKS, *PATCH[1:])),
        ("embed.conv_surface.bias", (C,)),
    ]
    for layer in (1, 2, 3, 4):
        c = g.dim(layer)
        heads = HEADS[layer - 1]
        for i in range(DEPTHS[layer - 1]):
            p = f"layer{layer}.block{i}."
            spec += [
                (p + "attn.bias_table", (BIAS_TABLE_ROWS, g.window_types(layer), heads)),
                (p + "attn.qkv.weight", (3 * c, c)),
                (p + "attn.qkv.bias", (3 * c,)),
                (p + "attn.proj.weight", (c, c)),
                (p + "attn.proj.bias", (c,)),
                (p + "norm1.weight", (c,)),
                (p + "norm1.bias", (c,)),
                (p + "mlp.fc1.weight", (4 * c, c)),
                (p + "mlp.fc1.bias", (4 * c,)),
                (p + "mlp.fc2.weight", (c, 4 * c)),
                (p + "mlp.fc2.bias", (c,)),
                (p + "norm2.weight", (c,)),
                (p + "norm2.bias", (c,)),
            ]
        if layer == 1:
            spec += [
                ("down.norm.weight", (4 * C,)),
                ("down.norm.bias", (4 * C,)),
                ("down.linear.weight", (2 * C, 4 * C)),
            ]
        if layer == 3:
            spec += [
                ("up.linear1.weight", (4 * C, 2 * C)),
                ("up.norm.weight", (C,)),
                ("up.norm.bias", (C,)),
                ("up.linear2.weight", (C, C)),
            ]
    spec += [
        ("recover.conv.weight", (2 * C, len(UPPER_VARS), *PATCH)),
        ("recover.conv.bias", (len(UPPER_VARS),)),
        ("recover.conv_surface.weight", (2 * C, len(SURFACE_VARS), *PATCH[1:])),
        ("recover.conv_surface.bias", (len(SURFACE_VARS),)),
    ]
    return spec


def param_offsets(g: PanguGeometry) -> tuple[dict[str, tuple[int, tuple[int, ...]]], int]:
    """name -> (element offset, shape) in the master blob; offsets are multiples of 64 floats."""
    out, off = {}, 0
    for name, shape in param_spec(g):
        n = 1
        for s in shape:
            n *= s
        out[name] = (off, shape)
        off += (n + 63) // 64 * 64
    return out, off


# ERA5-magnitude normalisation constants (mean, std) per channel, reference channel order.
_Z_MEAN = [740., 7.1e3, 1.38e4, 2.89e4, 4.07e4, 5.41e4, 6.97e4, 8.88e4, 1.005e5, 1.146e5, 1.32e5, 1.57e5, 1.99e5]
_Z_STD = [1.0e3, 9.5e2, 1.0e3, 1.4e3, 1.9e3, 2.5e3, 3.3e3, 4.4e3, 4.9e3, 5.3e3, 5.5e3, 5.4e3, 5.6e3]
_Q_MEAN = [9.0e-3, 7.5e-3, 6.0e-3, 3.3e-3, 1.5e-3, 8.5e-4, 3.9e-4, 1.3e-4, 6.0e-5, 2.0e-5, 5.0e-6, 2.7e-6, 2.7e-6]
_Q_STD = [5.9e-3, 5.2e-3, 4.2e-3, 2.6e-3, 1.6e-3, 1.1e-3, 5.2e-4, 1.7e-4, 8.0e-5, 2.5e-5, 4.0e-6, 6.0e-7, 3.0e-7]
_T_MEAN = [281., 277., 274., 267., 261., 253., 242., 229., 222., 218., 213., 208., 212.]
_T_STD = [17., 16., 15., 14., 13., 12.5, 12., 10.5, 8.5, 7., 8.5, 11., 9.]
_U_MEAN = [-0.03, 0.5, 1.4, 3.3, 4.9, 6.6, 8.9, 11.8, 13.4, 14.2, 13.5, 10.3, 5.4]
_U_STD = [6., 8., 8.2, 9.2, 10.5, 12., 14.5, 17.5, 18., 17., 15., 13., 14.]
_V_MEAN = [0.19, 0.2, 0.14, 0.02, -0.03, -0.03, -0.02, -0.02, -0.03, -0.04, -0.01, 0.01, 0.0]
_V_STD = [5.3, 6.2, 6.3, 7., 8., 9.2, 11., 13.5, 13.6, 12., 9.5, 7.3, 6.5]
_S_MEAN = [1.0096e5, -0.05, 0.19, 278.5]
_S_STD = [1.33e3, 5.5, 4.8, 21.3]


def channel_stats() -> tuple[torch.Tensor, torch.Tensor]:
    mean = _Z_MEAN + _Q_MEAN + _T_MEAN + _U_MEAN + _V_MEAN + _S_MEAN
    std = _Z_STD + _Q_STD + _T_STD + _U_STD + _V_STD + _S_STD
    return torch.tensor(mean, dtype=torch.float32), torch.tensor(std, dtype=torch.float32)


def init_synthetic(g: PanguGeometry, seed: int = 0) -> dict[str, torch.Tensor]:
    """Seeded random-init parameters of the right architecture (there is no network for the
    real checkpoint): weights/bias tables trunc-normal(0.02), biases N(0, 0.02),
    LayerNorm gamma = 1 + 0.05 N, beta = 0.05 N, constant masks U[0,1]."""
    gen = torch.Generator().manual_seed(seed)
    mean, std = channel_stats()
    params: dict[str, torch.Tensor] = {}
    for name, shape in param_spec(g):
        if name == "norm.mean":
            t = mean.clone()
        elif name == "norm.std":
            t = std.clone()
        elif name == "const_masks":
            t = torch.rand(shape, generator=gen)
        elif name.endswith("norm.weight") or name.endswith("norm1.weight") or name.endswith("norm2.weight"):
            t = 1.0 + 0.05 * torch.randn(shape, generator=gen)
        elif name.endswith("norm.bias") or name.endswith("norm1.bias") or name.endswith("norm2.bias"):
            t = 0.05 * torch.randn(shape, generator=gen)
        elif name.endswith(".bias"):
            t = 0.02 * torch.randn(shape, generator=gen)
        else:
            t = torch.empty(shape)
            torch.nn.init.trunc_normal_(t, std=0.02, a=-0.04, b=0.04, generator=gen)
        params[name] = t.float().contiguous()
    return params


def smooth_noise(g: PanguGeometry, seed: int) -> torch.Tensor:
    """(69, n_lat, n_lon) unit-variance fields: 9x9 box-smoothed N(0,1), periodic in longitude."""
    gen = torch.Generator().manual_seed(seed)
    n = torch.randn(g.n_channels, g.n_lat, g.n_lon, generator=gen)
    k = 9
    n = torch.nn.functional.pad(n[None], (k // 2, k // 2, 0, 0), mode="circular")
    n = torch.nn.functional.pad(n, (0, 0, k // 2, k // 2), mode="replicate")
    n = torch.nn.functional.avg_pool2d(n, k, stride=1)[0]
    return n / n.flatten(1).std(1)[:, None, None]


def synthetic_state(g: PanguGeometry, seed: int = 0, member: int | None = None) -> torch.Tensor:
    """(69, n_lat, n_lon) fp32 state: mean_c + std_c * (9x9 box-smoothed N(0,1), unit variance).
    ``member`` adds the ensemble perturbation 1e-3 * std_c * N(0,1; seed 1000+member)."""
    mean, std = channel_stats()
    n = smooth_noise(g, seed)
    x = mean[:, None, None] + std[:, None, None] * n
    if member is not None:
        g2 = torch.Generator().manual_seed(1000 + member)
        x = x + 1e-3 * std[:, None, None] * torch.randn(x.shape, generator=g2)
    return x.contiguous()
