"""CPU tests of the oracle (oracle/pangu_oracle.py): golden vectors, structure, invariances.

The golden vectors are SELF-ORACLE (see tests/golden/make_golden.py): reference parity is unpinned
because the reference's arithmetic is un-vendored third-party code."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import pangu_oracle as O
from skyrim_amd.pangu import spec

GOLD = Path(__file__).parent / "golden" / "pangu_toy_49x192.npz"


def test_geometry_full_grid_matches_survey():
    # 8x181x360 tokens, 8x91x180 after DownSample, 3720 / 960 windows, 124 / 64 types
    g = O.Geometry(721, 1440)
    assert g.res(1) == (8, 181, 360) and g.res(2) == (8, 91, 180)
    assert g.window_types(1) == 124 and g.window_types(2) == 64
    assert g.lat_pad == (724, 1, 2) and g.lev_pad == (14, 0, 1)
    p = spec.PanguGeometry(721, 1440)
    assert (p.H1, p.W1, p.H2, p.W2) == (181, 360, 91, 180)
    assert p.n_windows(1) == 3720 and p.n_windows(2) == 960
    assert p.window_types(1) == 124 and p.window_types(2) == 64
    n_params = sum(int(np.prod(s)) for n, s in spec.param_spec(p) if n != "const_masks" and not n.startswith("norm."))
    assert 63.5e6 < n_params < 64.8e6          # "64 M parameters"


def test_channel_order_matches_reference_wrapper():
    # /root/reference/skyrim/core/models/pangu.py:6-13
    assert spec.CHANNELS[:3] == ["z1000", "z925", "z850"] and spec.CHANNELS[13] == "q1000"
    assert spec.CHANNELS[-4:] == ["msl", "u10m", "v10m", "t2m"] and len(spec.CHANNELS) == 69


def test_position_index_structure():
    idx = O.position_index()
    assert idx.shape == (144, 144) and idx.min() == 0 and idx.max() == 3311
    # relative in longitude: shifting q and k by the same dw keeps the index
    assert torch.equal(idx[0:11, 0:11], idx[1:12, 1:12])
    # absolute in z / lat: different (h_q, h_k) pairs never share an index
    assert idx[0, 0] != idx[12, 12]
    gold = np.load(GOLD)
    assert np.array_equal(idx[::5, ::7].numpy(), gold["position_index_sub"])


def test_shift_mask_only_in_last_z_and_lat_windows():
    m = O.shifted_window_mask(8, 18, 48, torch.float32)        # (types=4*3, 1, 144, 144)
    assert m.shape == (12, 1, 144, 144)
    per_type = (m != 0).flatten(1).any(1)
    nH = 3
    for t in range(12):
        zi, hi = divmod(t, nH)
        assert bool(per_type[t]) == (zi == 3 or hi == nH - 1)
    assert set(m.unique().tolist()) == {-100.0, 0.0}


def test_golden_vectors(toy):
    g, params, x = toy
    gold = np.load(GOLD)
    assert np.allclose(x[:, ::6, ::16].numpy(), gold["state_in_sub"], rtol=1e-5, atol=1e-6)
    taps = {}
    y = O.forward(params, x.double(), taps=taps)
    y2 = O.forward(params, y)

    def close(a, b, tol=2e-5):
        a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
        assert np.abs(a - b).max() <= tol * np.abs(b).max(), np.abs(a - b).max() / np.abs(b).max()

    close(taps["embed"][::97, ::7], gold["embed_sub"])
    close(taps["layer1.block1"][::97, ::7], gold["layer1_block1_sub"])
    close(taps["down"][::31, ::11], gold["down_sub"])
    close(taps["layer3"][::31, ::11], gold["layer3_sub"])
    close(taps["up"][::97, ::7], gold["up_sub"])
    close(taps["layer4"][::97, ::7], gold["layer4_sub"])
    for c in range(69):   # per channel, like the metric
        close(y[c, ::6, ::16], gold["step1_sub"][c])
        close(y2[c, ::6, ::16], gold["step2_sub"][c], 1e-4)
    close(y.flatten(1).abs().max(1).values, gold["step1_channel_absmax"])


def test_fp32_matches_fp64(toy):
    g, params, x = toy
    e = O.per_channel_rel_err(O.forward(params, x), O.forward(params, x.double()))
    assert e.max() < 2e-5


def test_longitude_shift_equivariance(toy):
    """Longitude is periodic and the bias is relative in lon: shifting state and constant masks by a
    multiple of 96 columns (patch 4 x merge 2 x window 12) shifts the forecast by the same amount."""
    g, params, x = toy
    y = O.forward(params, x)
    p2 = dict(params)
    p2["const_masks"] = torch.roll(params["const_masks"], 96, dims=-1)
    y2 = O.forward(p2, torch.roll(x, 96, dims=-1))
    assert O.per_channel_rel_err(torch.roll(y2, -96, dims=-1), y).max() < 1e-4


def test_rollout_is_repeated_6h_step(toy):
    # GlobalModel.rollout re-instantiates the time loop every step (base.py:119-146): 6-h net each step
    g, params, x = toy
    outs = O.rollout(params, x, 2)
    assert torch.equal(outs[1], O.forward(params, outs[0]))


def test_switchable_conventions_roll_sign_mask_value_pad(toy):
    """The three conventions the pseudocode leaves open (oracle header; include/skyrim_pangu.h skpangu_config) each change the
    result, so a real ONNX file can discriminate them; the defaults reproduce the golden vectors (test_golden_vectors)."""
    g, params, x = toy
    base = O.forward(params, x)
    # +1: roll3D(x, shift=+half) first -> the FIRST Z / latitude window mixes wrapped rows and is the masked one
    plus = O.Conventions(roll_sign=+1)
    m = O.shifted_window_mask(8, 18, 48, torch.float32, plus)
    for t in range(12):
        zi, hi = divmod(t, 3)
        assert bool((m[t] != 0).any()) == (zi == 0 or hi == 0)
    xr = torch.arange(8 * 18 * 48.0).reshape(8, 18, 48)
    assert torch.roll(xr, (1, 3, 6), (0, 1, 2))[1, 3, 6] == xr[0, 0, 0]          # what "+1" means: rolled[p] = x[p - shift]
    assert set(O.shifted_window_mask(8, 18, 48, torch.float32, O.Conventions(mask_value=-1000.0)).unique().tolist()) == {-1000.0, 0.0}
    for conv in (plus, O.Conventions(pad="back"), O.Conventions(mask_value=-3.0)):
        y = O.forward(params, x, conv=conv)
        assert torch.isfinite(y).all() and O.per_channel_rel_err(y, base).max() > 1e-3, conv
    # -100 vs -1000 is numerically the same mask (exp underflows either way)
    assert O.per_channel_rel_err(O.forward(params, x, conv=O.Conventions(mask_value=-1000.0)), base).max() < 1e-5
    # surface slab position, qkv packing, bias index reading: each changes the result ...
    for conv in (O.Conventions(surface="last"), O.Conventions(qkv_order="h3d"), O.Conventions(bias_index="kq")):
        y = O.forward(params, x, conv=conv)
        assert torch.isfinite(y).all() and O.per_channel_rel_err(y, base).max() > 1e-3, conv
    # ... and the two that are pure re-labellings of parameters are EXACTLY the default network on re-labelled parameters:
    # qkv rows (3, heads, hd) -> (heads, 3, hd); bias table entries with query and key exchanged
    ph, pk = dict(params), dict(params)
    idx_qk = O.position_index()                                           # [q, k]
    swap = torch.empty(3312, dtype=torch.long)
    swap[idx_qk.reshape(-1)] = idx_qk.T.reshape(-1)                       # entry for (q, k) <- entry for (k, q); consistent: idx depends on the pair only
    assert torch.equal(swap[swap], torch.arange(3312))
    for name, w in params.items():
        if name.endswith("attn.qkv.weight") or name.endswith("attn.qkv.bias"):
            C3 = w.shape[0]
            heads = C3 // 3 // 32
            ph[name] = w.reshape(3, heads, 32, *w.shape[1:]).transpose(0, 1).reshape(w.shape)
        if name.endswith("attn.bias_table"):
            pk[name] = w[swap]
    assert torch.equal(O.forward(ph, x, conv=O.Conventions(qkv_order="h3d")), base)
    assert torch.allclose(O.forward(pk, x, conv=O.Conventions(bias_index="kq")), base, rtol=0, atol=0)
    gb = O.Geometry(721, 1440, "back")
    assert gb.lat_pad == (724, 0, 3) and gb.res(1) == (8, 181, 360) and spec.PanguGeometry(721, 1440, "back").pad_top(1) == 0
    assert spec.PanguGeometry(721, 1440).pad_top(1) == 2 and spec.PanguGeometry(721, 1440).lat_pad_top == 1
