"""CPU restatement (oracle) of the Pangu-Weather 6-h forward step.

TEST INFRASTRUCTURE ONLY.  Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this module; the product path
(``skyrim_amd``) never does and fails loudly without its HIP extension.

PARITY UNPINNED.  The arithmetic the reference runs for this path lives in
un-vendored third-party code: ``earth2mip.networks.pangu.load`` executing the
``pangu_weather_6.onnx`` graph under onnxruntime
(call site /root/reference/skyrim/core/models/pangu.py:45-46, contract doc
pangu.py:32-36, channel order pangu.py:6-13; requirements.txt:2 installs an
unpinned fork HEAD of earth2mip).  Neither earth2mip, onnxruntime nor the ONNX
weights exist in the build container, and the reference's own tests hold no
numerical vector for this path (tests/core/test_skyrim.py:6-10 asserts an exit
code only).  This file therefore restates the *published* algorithm -- Bi et
al. 2023, "Accurate medium-range global weather forecasting with 3D neural
networks", and the authors' public pseudocode (198808xc/Pangu-Weather,
pseudocode.py) -- in plain PyTorch-CPU fp32/fp64, written independently of the
HIP kernels.  Golden vectors under tests/golden/ are produced by THIS file
("self-oracle"); see DESIGN.md.

Conventions the pseudocode leaves open.  The three that a real ``pangu_weather_6.onnx`` could
settle either way are SWITCHABLE (``Conventions``; the engine takes the same three through
``skpangu_config``), each default with its source:
  * ``pad``: "centre" (default) = front = total // 2, back = the rest (13->14 levels: 0/1;
    721->724 lat: 1/2; 181->186 tokens: 2/3; 181->182: 0/1) -- the pseudocode only says
    "zero-pad"; public re-implementations centre.  "back" puts all of it at the end.
  * ``roll_sign``: -1 (default) = shifted windows roll by -(wz//2, wh//2, ww//2) = -(1, 3, 6),
    the Swin convention; +1 = ``roll3D(x, shift=[+1, +3, +6])`` as the pseudocode's call is
    literally written (then -half to restore).  The mask regions follow the sign (the window
    that mixes wrapped and unwrapped rows is the last one for -1, the first one for +1).
  * ``mask_value``: -100 (default, Swin) -- the pseudocode's comment suggests -1000.
    The mask is generated over Z and lat only: longitude is periodic, so rolled lon windows
    are genuine neighbours
  * the surface slab is token level 0, the 7 upper-air slabs follow
    (PatchRecovery reads x[:, :, 0] as surface, x[:, :, 1:] as upper air)
  * the network sees (x - mean_c) / std_c and its output is de-normalised with the
    same per-channel constants (the ONNX graph carries them as initialisers)
  * blocks are post-norm:  x = x + LN(attn(x));  x = x + LN(mlp(x))
  * GELU is the exact erf form; LayerNorm eps = 1e-5; drop-path/dropout are
    identity at inference.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F

WINDOW = (2, 6, 12)
PATCH = (2, 4, 4)
N_LEVELS = 13
N_UPPER_VARS = 5      # z, q, t, u, v   (reference channel order pangu.py:6-13)
N_SURF_VARS = 4       # msl, u10m, v10m, t2m
N_CONST_MASKS = 3     # land mask, soil type, topography
DIM = 192
HEADS = (6, 12, 12, 6)
DEPTHS = (2, 6, 6, 2)
LN_EPS = 1e-5
MASK_VALUE = -100.0


@dataclass(frozen=True)
class Conventions:
    pad: str = "centre"          # "centre" | "back"
    roll_sign: int = -1          # -1: torch.roll(x, -half) first (Swin);  +1: torch.roll(x, +half) first (pseudocode as written)
    mask_value: float = MASK_VALUE
    # three more points the public pseudocode does not settle (each a prepare-time permutation in the engine, skpangu_config):
    surface: str = "first"       # token level of the surface slab: "first" (PatchRecovery reads the surface at index 0) | "last"
                                 # (PatchEmbedding's concatenate((input, input_surface)) taken literally: upper-air levels first)
    qkv_order: str = "3hd"       # packing of the qkv Linear's 3C outputs: (3, heads, head_dim) | "h3d" = (heads, 3, head_dim)
    bias_index: str = "qk"       # the (144 x 144) earth-specific bias gathered as [query][key] | "kq" = [key][query] (the transposed reading
                                 # of position_index: which of its two meshgrid axes is the query)


DEFAULT = Conventions()


def _centre_pad(n: int, mult: int, pad: str = "centre") -> tuple[int, int, int]:
    """(padded size, front, back) so that padded is the next multiple of mult."""
    padded = (n + mult - 1) // mult * mult
    total = padded - n
    front = total // 2 if pad == "centre" else 0
    return padded, front, total - front


@dataclass(frozen=True)
class Geometry:
    """Derived sizes for a (n_lat, n_lon) grid with 13 pressure levels."""
    n_lat: int
    n_lon: int
    pad: str = "centre"

    @property
    def lat_pad(self):          # input latitude -> multiple of 4
        return _centre_pad(self.n_lat, PATCH[1], self.pad)

    @property
    def lev_pad(self):          # 13 -> 14
        return _centre_pad(N_LEVELS, PATCH[0], self.pad)

    @property
    def Z(self):                # token levels: surface + 7 upper
        return 1 + self.lev_pad[0] // PATCH[0]

    @property
    def H1(self):
        return self.lat_pad[0] // PATCH[1]

    @property
    def W1(self):
        assert self.n_lon % (PATCH[2] * 2 * WINDOW[2]) == 0, "n_lon must be a multiple of 96"
        return self.n_lon // PATCH[2]

    @property
    def H2(self):
        return _centre_pad(self.H1, 2, self.pad)[0] // 2

    @property
    def W2(self):
        return self.W1 // 2

    def res(self, layer: int) -> tuple[int, int, int]:
        """(Z, H, W) token resolution of layer 1..4."""
        return (self.Z, self.H1, self.W1) if layer in (1, 4) else (self.Z, self.H2, self.W2)

    def window_types(self, layer: int) -> int:
        Z, H, W = self.res(layer)
        Hp = _centre_pad(H, WINDOW[1], self.pad)[0]
        return (Z // WINDOW[0]) * (Hp // WINDOW[1])


# --------------------------------------------------------------------------- #
#  building blocks
# --------------------------------------------------------------------------- #
def _emulate(t: torch.Tensor, emu):
    """Optionally round a GEMM operand to a 16-bit type (precision studies only)."""
    if emu is None:
        return t
    return t.to(emu).to(t.dtype)


def _linear(x, w, b=None, emu=None):
    return F.linear(_emulate(x, emu), _emulate(w, emu), b)


def position_index() -> torch.Tensor:
    """Earth-specific bias index, (144, 144) long; [q, k] (pseudocode _construct_index)."""
    wz, wh, ww = WINDOW
    coords_zi = torch.arange(wz)
    coords_zj = -torch.arange(wz) * wz
    coords_hi = torch.arange(wh)
    coords_hj = -torch.arange(wh) * wh
    coords_w = torch.arange(ww)
    coords_1 = torch.stack(torch.meshgrid(coords_zi, coords_hi, coords_w, indexing="ij"))
    coords_2 = torch.stack(torch.meshgrid(coords_zj, coords_hj, coords_w, indexing="ij"))
    f1 = coords_1.flatten(1)
    f2 = coords_2.flatten(1)
    coords = f1[:, :, None] - f2[:, None, :]
    coords = coords.permute(1, 2, 0).contiguous()
    coords[:, :, 2] += ww - 1
    coords[:, :, 1] *= 2 * ww - 1
    coords[:, :, 0] *= (2 * ww - 1) * wh * wh
    return coords.sum(-1)


def shifted_window_mask(Z: int, Hp: int, Wp: int, dtype, conv: Conventions = DEFAULT) -> torch.Tensor:
    """(nZ*nH, 1(nW), 144, 144) additive mask for rolled windows; Z and lat only."""
    wz, wh, ww = WINDOW
    sz, sh = wz // 2, wh // 2
    img = torch.zeros(Z, Hp, Wp)
    cnt = 0
    if conv.roll_sign < 0:       # rolled by -half: the LAST window holds [unwrapped | wrapped] rows
        z_regions = (slice(0, -wz), slice(-wz, -sz), slice(-sz, None))
        h_regions = (slice(0, -wh), slice(-wh, -sh), slice(-sh, None))
    else:                        # rolled by +half: the FIRST window holds [wrapped | unwrapped] rows
        z_regions = (slice(0, sz), slice(sz, wz), slice(wz, None))
        h_regions = (slice(0, sh), slice(sh, wh), slice(wh, None))
    for zs in z_regions:
        for hs in h_regions:
            img[zs, hs, :] = cnt
            cnt += 1
    win = img.reshape(Z // wz, wz, Hp // wh, wh, Wp // ww, ww)
    win = win.permute(0, 2, 4, 1, 3, 5).reshape(Z // wz * (Hp // wh), Wp // ww, wz * wh * ww)
    diff = win[:, :, None, :] - win[:, :, :, None]
    mask = torch.where(diff != 0, torch.tensor(float(conv.mask_value)), torch.tensor(0.0)).to(dtype)
    # identical for every longitude window: keep one
    return mask[:, :1]


def earth_attention(p, x_win, n_types, heads, mask, emu=None, conv: Conventions = DEFAULT):
    """x_win: (types, nW, 144, C) -> same shape.  EarthAttention3D.forward."""
    T, nW, L, C = x_win.shape
    hd = C // heads
    scale = hd ** -0.5
    qkv = _linear(x_win, p["attn.qkv.weight"], p["attn.qkv.bias"], emu)
    if conv.qkv_order == "3hd":
        qkv = qkv.reshape(T, nW, L, 3, heads, hd).permute(3, 0, 1, 4, 2, 5)
    else:                                                             # "h3d": (heads, 3, head_dim)
        qkv = qkv.reshape(T, nW, L, heads, 3, hd).permute(4, 0, 1, 3, 2, 5)
    q, k, v = qkv[0] * scale, qkv[1], qkv[2]
    att = _emulate(q, emu) @ _emulate(k, emu).transpose(-1, -2)      # (T, nW, heads, L, L)
    idx = position_index().reshape(-1)
    bias = p["attn.bias_table"][idx]                                  # (L*L, T, heads)
    bias = bias.reshape(L, L, n_types, heads)
    bias = bias.permute(2, 3, 0, 1) if conv.bias_index == "qk" else bias.permute(2, 3, 1, 0)      # (T, heads, query, key)
    att = att + bias[:, None]
    if mask is not None:
        att = att + mask[:, :, None]
    att = torch.softmax(att, dim=-1)
    out = _emulate(att, emu) @ _emulate(v, emu)                        # (T, nW, heads, L, hd)
    out = out.permute(0, 1, 3, 2, 4).reshape(T, nW, L, C)
    return _linear(out, p["attn.proj.weight"], p["attn.proj.bias"], emu)


def earth_block(p, x, res, heads, roll, emu=None, conv: Conventions = DEFAULT):
    """One EarthSpecificBlock.  x: (Z*H*W, C)."""
    Z, H, W = res
    wz, wh, ww = WINDOW
    C = x.shape[-1]
    shortcut = x
    x = x.reshape(Z, H, W, C)
    Hp, top, bot = _centre_pad(H, wh, conv.pad)
    x = F.pad(x, (0, 0, 0, 0, top, bot))                               # pad lat only (Z, W already fit)
    sg = 1 if conv.roll_sign > 0 else -1
    if roll:
        x = torch.roll(x, shifts=(sg * (wz // 2), sg * (wh // 2), sg * (ww // 2)), dims=(0, 1, 2))
        mask = shifted_window_mask(Z, Hp, W, x.dtype, conv)
    else:
        mask = None
    nZ, nH, nW = Z // wz, Hp // wh, W // ww
    xw = x.reshape(nZ, wz, nH, wh, nW, ww, C).permute(0, 2, 4, 1, 3, 5, 6)
    xw = xw.reshape(nZ * nH, nW, wz * wh * ww, C)
    xw = earth_attention(p, xw, nZ * nH, heads, mask, emu, conv)
    x = xw.reshape(nZ, nH, nW, wz, wh, ww, C).permute(0, 3, 1, 4, 2, 5, 6).reshape(Z, Hp, W, C)
    if roll:
        x = torch.roll(x, shifts=(-sg * (wz // 2), -sg * (wh // 2), -sg * (ww // 2)), dims=(0, 1, 2))
    x = x[:, top:top + H].reshape(Z * H * W, C)
    x = shortcut + F.layer_norm(x, (C,), p["norm1.weight"], p["norm1.bias"], LN_EPS)
    h = _linear(x, p["mlp.fc1.weight"], p["mlp.fc1.bias"], emu)
    h = F.gelu(h)
    h = _linear(h, p["mlp.fc2.weight"], p["mlp.fc2.bias"], emu)
    return x + F.layer_norm(h, (C,), p["norm2.weight"], p["norm2.bias"], LN_EPS)


def patch_embed(p, g: Geometry, upper, surface, emu=None, conv: Conventions = DEFAULT):
    """upper (5,13,H,W), surface (4,H,W), both already normalised -> (Z*H1*W1, 192)."""
    _, lt, lb = g.lat_pad
    _, zf, zb = g.lev_pad
    upper = F.pad(upper, (0, 0, lt, lb, zf, zb))
    surf = torch.cat([surface, p["const_masks"].to(surface.dtype)], 0)
    surf = F.pad(surf, (0, 0, lt, lb))
    xu = F.conv3d(_emulate(upper, emu)[None], _emulate(p["embed.conv.weight"], emu),
                  p["embed.conv.bias"], stride=PATCH)[0]              # (C, 7, H1, W1)
    xs = F.conv2d(_emulate(surf, emu)[None], _emulate(p["embed.conv_surface.weight"], emu),
                  p["embed.conv_surface.bias"], stride=PATCH[1:])[0]   # (C, H1, W1)
    x = torch.cat([xs[:, None], xu], 1) if conv.surface == "first" else torch.cat([xu, xs[:, None]], 1)      # where the surface slab sits
    return x.permute(1, 2, 3, 0).reshape(-1, x.shape[0])


def downsample(p, g: Geometry, x, emu=None):
    Z, H, W = g.res(1)
    C = x.shape[-1]
    x = x.reshape(Z, H, W, C)
    He, top, bot = _centre_pad(H, 2, g.pad)
    x = F.pad(x, (0, 0, 0, 0, top, bot))
    x = x.reshape(Z, He // 2, 2, W // 2, 2, C).permute(0, 1, 3, 2, 4, 5).reshape(-1, 4 * C)
    x = F.layer_norm(x, (4 * C,), p["down.norm.weight"], p["down.norm.bias"], LN_EPS)
    return _linear(x, p["down.linear.weight"], None, emu)


def upsample(p, g: Geometry, x, emu=None):
    Z, H2, W2 = g.res(2)
    _, H1, W1 = g.res(1)
    x = _linear(x, p["up.linear1.weight"], None, emu)                  # (N2, 4*C_out)
    C = x.shape[-1] // 4
    x = x.reshape(Z, H2, W2, 2, 2, C).permute(0, 1, 3, 2, 4, 5).reshape(Z, 2 * H2, 2 * W2, C)
    _, top, _ = _centre_pad(H1, 2, g.pad)
    x = x[:, top:top + H1, :W1].reshape(-1, C)
    x = F.layer_norm(x, (C,), p["up.norm.weight"], p["up.norm.bias"], LN_EPS)
    return _linear(x, p["up.linear2.weight"], None, emu)


def patch_recover(p, g: Geometry, x, emu=None, conv: Conventions = DEFAULT):
    """x (Z*H1*W1, 384) -> upper (5,13,H,W), surface (4,H,W) (still normalised)."""
    Z, H1, W1 = g.res(1)
    C = x.shape[-1]
    x = x.reshape(Z, H1, W1, C).permute(3, 0, 1, 2)
    xu, xs = (x[:, 1:], x[:, 0]) if conv.surface == "first" else (x[:, :-1], x[:, -1])
    up = F.conv_transpose3d(_emulate(xu, emu)[None], _emulate(p["recover.conv.weight"], emu),
                            p["recover.conv.bias"], stride=PATCH)[0]
    sf = F.conv_transpose2d(_emulate(xs, emu)[None], _emulate(p["recover.conv_surface.weight"], emu),
                            p["recover.conv_surface.bias"], stride=PATCH[1:])[0]
    _, lt, _ = g.lat_pad
    _, zf, _ = g.lev_pad
    up = up[:, zf:zf + N_LEVELS, lt:lt + g.n_lat, :g.n_lon]
    sf = sf[:, lt:lt + g.n_lat, :g.n_lon]
    return up, sf


def _block_params(params, layer, i):
    pre = f"layer{layer}.block{i}."
    return {k[len(pre):]: v for k, v in params.items() if k.startswith(pre)}


def split_state(x):
    """(69,H,W) in reference channel order -> upper (5,13,H,W), surface (4,H,W)."""
    nu = N_UPPER_VARS * N_LEVELS
    return x[:nu].reshape(N_UPPER_VARS, N_LEVELS, *x.shape[1:]), x[nu:]


def forward(params: dict, x: torch.Tensor, emu=None, taps: dict | None = None, conv: Conventions = DEFAULT) -> torch.Tensor:
    """One 6-h step.  x: (69, n_lat, n_lon) physical units -> same shape.

    ``taps`` (optional dict) receives intermediate activations for kernel-level tests; ``conv`` selects the open conventions.
    """
    dt = x.dtype
    params = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in params.items()}
    g = Geometry(x.shape[1], x.shape[2], conv.pad)
    mean = params["norm.mean"][:, None, None]
    std = params["norm.std"][:, None, None]
    xn = (x - mean) / std
    upper, surface = split_state(xn)
    t = patch_embed(params, g, upper, surface, emu, conv)
    if taps is not None:
        taps["embed"] = t
    for i in range(DEPTHS[0]):
        t = earth_block(_block_params(params, 1, i), t, g.res(1), HEADS[0], i % 2 == 1, emu, conv)
        if taps is not None:
            taps[f"layer1.block{i}"] = t
    skip = t
    t = downsample(params, g, t, emu)
    if taps is not None:
        taps["down"] = t
    for layer in (2, 3):
        for i in range(DEPTHS[layer - 1]):
            t = earth_block(_block_params(params, layer, i), t, g.res(layer), HEADS[layer - 1], i % 2 == 1, emu, conv)
            if taps is not None:
                taps[f"layer{layer}.block{i}"] = t
    if taps is not None:
        taps["layer3"] = t
    t = upsample(params, g, t, emu)
    if taps is not None:
        taps["up"] = t
    for i in range(DEPTHS[3]):
        t = earth_block(_block_params(params, 4, i), t, g.res(4), HEADS[3], i % 2 == 1, emu, conv)
        if taps is not None:
            taps[f"layer4.block{i}"] = t
    if taps is not None:
        taps["layer4"] = t
    t = torch.cat([skip, t], -1)
    up, sf = patch_recover(params, g, t, emu, conv)
    y = torch.cat([up.reshape(-1, *up.shape[2:]), sf], 0)
    return y * std + mean


def rollout(params: dict, x: torch.Tensor, n_steps: int, emu=None, conv: Conventions = DEFAULT):
    """Autoregressive rollout with the 6-h network on every step -- the behaviour of
    GlobalModel.rollout (/root/reference/skyrim/core/models/base.py:119-146), which
    re-instantiates the time loop each step."""
    outs = []
    for _ in range(n_steps):
        x = forward(params, x, emu, conv=conv)
        outs.append(x)
    return outs


def per_channel_rel_err(y: torch.Tensor, ref: torch.Tensor) -> torch.Tensor:
    """max|y - ref| / max|ref| per channel."""
    num = (y.double() - ref.double()).abs().flatten(1).max(1).values
    den = ref.double().abs().flatten(1).max(1).values
    return num / den


def per_channel_sigma_err(y: torch.Tensor, ref: torch.Tensor, std: torch.Tensor) -> torch.Tensor:
    """The same difference in the network's own units: max|y - ref| / sigma_c with sigma_c the channel's normalisation constant
    (params["norm.std"]).  max|ref| flatters channels that sit on a large offset (msl: mean 1.0e5 Pa, sigma 1.3e3 Pa -- the 1e-3 bar of
    ``per_channel_rel_err`` is 7.5 % of sigma there); this figure does not.  Reported beside the SURVEY.md 8(d) metric, never instead of it."""
    return (y - ref).abs().flatten(1).max(1).values / std.reshape(-1).to(y.dtype)

