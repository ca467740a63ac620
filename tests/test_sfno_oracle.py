"""CPU tests of the SFNO row: the oracle's spherical-harmonic transform against closed-form identities and the golden
fixture, the product-side transform matrices (skyrim_amd/sfno/sht.py) against the oracle, and the C ABI of
libskyrim_sfno.so (symbols + argument errors; no compute without a GPU)."""
import ctypes
import math
import re
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import sfno_oracle as O
from skyrim_amd.sfno import engine as E
from skyrim_amd.sfno.sht import ShtMatrices, colatitudes_and_weights
from skyrim_amd.sfno.spec import CHANNELS, SfnoConfig, flops_per_step, init_synthetic, param_spec, synthetic_state

TINY = SfnoConfig(n_lat=33, n_lon=64, in_chans=5, out_chans=5, embed_dim=16, num_layers=3, scale_factor=2)
GOLD = Path(__file__).resolve().parent / "golden" / "sfno_tiny_33x64.npz"


@pytest.mark.parametrize("grid,n_lat,n_lon,lmax", [("legendre-gauss", 16, 32, 16), ("equiangular", 33, 64, 16), ("equiangular", 49, 96, 24)])
def test_sht_closed_forms_and_round_trip(grid, n_lat, n_lon, lmax):
    mmax = min(n_lon // 2, lmax)
    t = O.SHT(n_lat, n_lon, lmax, mmax, grid)
    theta = (O.legendre_gauss(n_lat) if grid == "legendre-gauss" else O.clenshaw_curtis(n_lat))[0]
    one = t.forward(torch.ones(n_lat, n_lon, dtype=torch.float64))
    assert abs(one[0, 0].real.item() - math.sqrt(4 * math.pi)) < 1e-12 and (one.abs().sum() - one[0, 0].abs()).item() < 1e-10
    lon = torch.arange(n_lon, dtype=torch.float64) * (2 * math.pi / n_lon)
    th = torch.from_numpy(theta)[:, None]
    # Y_1^0 = sqrt(3/4pi) cos(theta);  Re Y_1^1 = -sqrt(3/8pi) sin(theta) cos(lon)  (Condon-Shortley)
    c = t.forward(torch.cos(th).expand(n_lat, n_lon).contiguous())
    assert abs(c[1, 0].real.item() - math.sqrt(4 * math.pi / 3)) < 1e-12
    c = t.forward(torch.sin(th) * torch.cos(lon)[None, :])
    assert abs(c[1, 1].real.item() + math.sqrt(2 * math.pi / 3)) < 1e-12 and abs(c[1, 1].imag.item()) < 1e-12
    # synthesis then analysis of a band-limited field is the identity (quadrature exact below the grid's degree)
    gen = torch.Generator().manual_seed(1)
    co = torch.complex(torch.randn(2, lmax, mmax, generator=gen, dtype=torch.float64), torch.randn(2, lmax, mmax, generator=gen, dtype=torch.float64))
    co = co * (torch.arange(lmax)[:, None] >= torch.arange(mmax)[None, :])
    co[..., 0] = co[..., 0].real + 0j
    if grid == "equiangular":
        co[:, lmax // 2:] = 0
    assert (t.forward(t.inverse(co)) - co).abs().max().item() < 1e-12


def test_quadrature_weights_integrate_polynomials():
    for grid, n in (("equiangular", 33), ("legendre-gauss", 16)):
        theta, w = colatitudes_and_weights(n, grid)
        x = np.cos(theta)
        assert abs(w.sum() - 2.0) < 1e-13 and abs((w * x ** 2).sum() - 2.0 / 3.0) < 1e-13 and abs((w * x ** 5).sum()) < 1e-13
        t2, w2 = (O.legendre_gauss(n) if grid == "legendre-gauss" else O.clenshaw_curtis(n))
        assert np.allclose(theta, t2, atol=1e-14) and np.allclose(w, w2, atol=1e-14)        # two derivations, one rule


@pytest.mark.parametrize("grid,n_lat,n_lon,lmax", [("equiangular", 33, 64, 16), ("legendre-gauss", 16, 32, 16)])
def test_product_transform_matrices_match_the_oracle(grid, n_lat, n_lon, lmax):
    mmax = min(n_lon // 2, lmax)
    m, o = ShtMatrices(n_lat, n_lon, lmax, mmax, grid), O.SHT(n_lat, n_lon, lmax, mmax, grid)
    x = torch.randn(3, n_lat, n_lon, dtype=torch.float64, generator=torch.Generator().manual_seed(2))
    ref = o.forward(x)
    f = torch.einsum("clj,nj->cln", x, torch.from_numpy(m.dft).double())
    coef = torch.einsum("ckm,mlk->clm", torch.complex(f[..., 0::2], f[..., 1::2]), torch.from_numpy(m.analysis).double().to(torch.complex128))
    assert (coef - ref).abs().max().item() < 2e-7 * ref.abs().max().item()
    g = torch.einsum("clm,mkl->ckm", ref, torch.from_numpy(m.synthesis).double().to(torch.complex128))
    xr = torch.einsum("ckn,jn->ckj", torch.stack([g.real, g.imag], -1).reshape(3, n_lat, 2 * mmax), torch.from_numpy(m.idft).double())
    assert (xr - o.inverse(ref)).abs().max().item() < 5e-7 * x.abs().max().item()
    with pytest.raises(ValueError):
        ShtMatrices(n_lat, n_lon, lmax, n_lon // 2 + 1, grid)


def test_oracle_matches_golden_fixture_and_is_deterministic():
    gold = np.load(GOLD)
    params, x = init_synthetic(TINY, 0), synthetic_state(TINY, 0)
    assert np.array_equal(x.numpy(), gold["state_in"])
    taps = {}
    y = O.forward(params, x, TINY, taps=taps)
    assert np.allclose(y.numpy(), gold["step1"], rtol=0, atol=2e-5 * np.abs(gold["step1"]).max())
    assert np.allclose(O.forward(params, y, TINY).numpy(), gold["step2"], rtol=0, atol=5e-5 * np.abs(gold["step2"]).max())
    assert np.allclose(taps["encoder"][::3, ::4, ::8].numpy(), gold["encoder_sub"], atol=1e-5)
    assert np.allclose(taps["blocks.0.out"][::3, ::2, ::4].numpy(), gold["block0_out_sub"], atol=1e-4)
    assert torch.equal(y, O.forward(params, x, TINY))
    coef = O.SHT(TINY.n_lat, TINY.n_lon, TINY.lmax, TINY.mmax, "equiangular").forward(x[:2].double())
    assert np.allclose(coef.real.numpy(), gold["sht_coef_re"], atol=1e-3) and np.allclose(coef.imag.numpy(), gold["sht_coef_im"], atol=1e-3)


def test_spec_shapes_and_flops():
    full = SfnoConfig()
    assert (full.h, full.w, full.lmax, full.mmax) == (240, 480, 240, 240) and len(CHANNELS) == 73
    assert CHANNELS[:8] == ["u10m", "v10m", "u100m", "v100m", "t2m", "sp", "msl", "tcwv"] and CHANNELS[-1] == "r1000" and CHANNELS[8] == "u50"
    names = [n for n, _ in param_spec(full)]
    assert len(names) == len(set(names)) == 6 + 11 * 8 + 3
    assert dict(param_spec(full))["blocks.3.filter.weight"] == (256, 256, 240, 2)
    assert 1.5e12 < flops_per_step(full) < 2.5e12
    p = init_synthetic(TINY, 0)
    assert all(tuple(p[n].shape) == s for n, s in param_spec(TINY))


def test_sfno_library_exports_declared_symbols_and_rejects_bad_arguments():
    header = (Path(__file__).resolve().parent.parent / "include" / "skyrim_sfno.h").read_text()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    syms = sorted(set(re.findall(r"\b(sksfno_[a-z_]+)\s*\(", header)))
    lib = E.load_library()
    assert set(syms) == set(E.EXPORTS) and all(hasattr(lib, s) for s in syms)
    assert lib.sksfno_abi_version() == 2
    assert lib.sksfno_gemm_run(None, None) == -1
    d = E.GemmDesc()
    assert lib.sksfno_gemm_run(ctypes.byref(d), None) == -1                       # null pointers / zero sizes
    assert lib.sksfno_instance_norm(None, None, None, None, 4, 16, 1e-6, None) == -1
    assert lib.sksfno_prepare_weight(None, 1, 1, 4, 4, None, 0, 8, None) == -1
    # ABI v2: fused pixel-wise chains
    assert E.chain_shapes(lib) == [(256, 512, 96, 96), (64, 96, 32, 32)]
    assert lib.sksfno_chain_run(None, None) == -1
    assert lib.sksfno_chain_run(ctypes.byref(E.ChainDesc()), None) == -1
    one = ctypes.c_void_p(16)                                                     # non-null, never dereferenced: argument checks come first
    bad_hw = E.ChainDesc(E.CHAIN_MLP, 1, one, None, one, one, 24, 40, 11, 9, one, one, None, None, one)      # HW not a multiple of 16
    too_wide = E.ChainDesc(E.CHAIN_MLP, 1, one, None, one, one, 32, 65, 11, 9, one, one, None, None, one)    # embed wider than the class
    no_dec = E.ChainDesc(E.CHAIN_TAIL, 1, one, one, one, one, 32, 40, 11, 9, one, one, None, None, one)      # TAIL without decoder weights
    for d in (bad_hw, too_wide, no_dec):
        assert lib.sksfno_chain_run(ctypes.byref(d), None) == -1
    assert lib.sksfno_instance_stats(None, None, None, None, None, 4, 16, 1e-6, None) == -1
    assert lib.sksfno_prepare_chain_weights(one, one, 40, 64, 64, one, one, None) == -1                      # K not a multiple of 32


@pytest.mark.parametrize("grid,n_lat,n_lon,lmax", [("equiangular", 33, 64, 16), ("legendre-gauss", 24, 48, 24), ("equiangular", 97, 192, 32)])
def test_sht_pinned_against_scipy_spherical_harmonics(grid, n_lat, n_lon, lmax):
    """An INDEPENDENT implementation pins the transform conventions: scipy.special's spherical harmonics (orthonormal, Condon-Shortley
    phase) and associated Legendre functions.  (1) the oracle's / the product's Legendre tables equal sqrt((2l+1)/(4 pi) (l-m)!/(l+m)!)
    P_l^m; (2) analysing a field built from scipy's Y_l^m returns exactly its coefficients (torch-harmonics' convention: the m >= 0
    coefficients of a real field f = sum_l [c_l0 Y_l0 + 2 Re sum_{m>0} c_lm Y_lm]); (3) synthesis gives the field back."""
    from scipy import special
    mmax = min(n_lon // 2, lmax)
    theta = (O.legendre_gauss(n_lat) if grid == "legendre-gauss" else O.clenshaw_curtis(n_lat))[0]
    lon = np.arange(n_lon) * (2 * np.pi / n_lon)
    # (1) Legendre tables
    tab_o = O.legendre_ortho(mmax, lmax, theta)
    from skyrim_amd.sfno.sht import legendre_functions
    tab_p = legendre_functions(mmax, lmax, theta)
    for m in range(mmax):
        for l in range(m, lmax):
            norm = math.sqrt((2 * l + 1) / (4 * math.pi) * math.exp(special.gammaln(l - m + 1) - special.gammaln(l + m + 1)))
            want = norm * special.lpmv(m, l, np.cos(theta))
            assert np.allclose(tab_o[m, l], want, atol=1e-11) and np.allclose(tab_p[m, l], want, atol=1e-11), (m, l)
    # (2) + (3): a real field from scipy's complex harmonics, below the grid's exact-quadrature degree
    lcut = lmax if grid == "legendre-gauss" else lmax // 2
    gen = np.random.default_rng(4)
    c = (gen.standard_normal((lmax, mmax)) + 1j * gen.standard_normal((lmax, mmax))) * (np.arange(lmax)[:, None] >= np.arange(mmax)[None, :])
    c[lcut:] = 0
    c[:, 0] = c[:, 0].real
    sph = getattr(special, "sph_harm_y", None)
    field = np.zeros((n_lat, n_lon))
    for l in range(lcut):
        for m in range(min(l, mmax - 1) + 1):
            y = sph(l, m, theta[:, None], lon[None, :]) if sph is not None else special.sph_harm(m, l, lon[None, :], theta[:, None])
            field += (c[l, m] * y).real * (1.0 if m == 0 else 2.0)
    t = O.SHT(n_lat, n_lon, lmax, mmax, grid)
    got = t.forward(torch.from_numpy(field))
    assert np.abs(got.numpy() - c).max() < 1e-10
    assert np.abs(t.inverse(torch.from_numpy(c)).numpy() - field).max() < 1e-10
    # the product's GEMM matrices reproduce the same coefficients (fp32 tables: 1e-6)
    m_ = ShtMatrices(n_lat, n_lon, lmax, mmax, grid)
    f = field @ m_.dft.astype(np.float64).T                                                  # [lat][2m]
    coef = np.einsum("km,mlk->lm", f[:, 0::2] + 1j * f[:, 1::2], m_.analysis.astype(np.float64))
    assert np.abs(coef - c).max() < 2e-6 * np.abs(c).max()


@pytest.mark.parametrize("grid,n_lat,n_lon", [("equiangular", 33, 64), ("legendre-gauss", 16, 32)])
def test_channel_mix_commutes_with_the_synthesis(grid, n_lat, n_lon):
    """What SfnoEngine relies on for the grid-changing blocks: a 1x1 convolution of a band-limited field is the synthesis
    of the channel-mixed coefficients, skip(iSHT(coef)) = iSHT(W coef), and a constant field b is the (l, m) = (0, 0) coefficient
    b sqrt(4 pi) -- checked in float64 on the oracle's own transforms (the engine folds W into the block's dhconv matrices)."""
    lmax = mmax = 16
    sht = O.SHT(n_lat, n_lon, lmax, mmax, grid, torch.float64)
    gen = torch.Generator().manual_seed(0)
    C = 6
    x = torch.randn(C, n_lat, n_lon, generator=gen, dtype=torch.float64)
    coef = sht.forward(x)                                                            # (C, L, M) complex
    w = torch.randn(C, C, generator=gen, dtype=torch.float64)
    b = torch.randn(C, generator=gen, dtype=torch.float64)
    grid_space = torch.einsum("oc,chw->ohw", w, sht.inverse(coef)) + b[:, None, None]
    mixed = torch.einsum("oc,clm->olm", w.to(coef.dtype), coef)
    mixed[:, 0, 0] += b * (4.0 * torch.pi) ** 0.5
    assert (sht.inverse(mixed) - grid_space).abs().max().item() < 1e-11
