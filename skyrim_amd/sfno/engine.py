"""SFNO (FourCastNet v2-small) 6-h step on one MI355X: the host owns buffers and call order, every FLOP runs in the
HIP kernels of include/skyrim_sfno.h (libskyrim_sfno.so, loaded through ctypes; PyTorch is device memory + streams).

One step = 88 launches at the default depth (8 blocks).  Data layouts (all fp32):

    activations            [C][H][W]                      (the reference's NCHW without the batch)
    longitude spectrum     [mmax][re, im][C][H padded]    truncated real DFT, m < mmax; order-major so that the Legendre
                                                          GEMM of one order reads / writes contiguous latitudes
    SH coefficients        [l][m][re, im][C]              (re/im, channel) last so that the per-degree complex channel mixing
                                                          is a plain GEMM over k = (re/im, channel)

The kernels read the raw state and write physical units: the input normalisation is a per-channel affine in the loader of
the two GEMMs that read the state (before the fp16 split: raw geopotential / pressure exceed the fp16 range), the output
de-normalisation is folded into the decoder's last matrix, the big-skip concat is two GEMMs into one accumulator chain.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes
import os
from pathlib import Path

import numpy as np
import torch

from .. import ops
from .sht import ShtMatrices
from .spec import SfnoConfig, param_spec

_LIB_PATH = Path(__file__).resolve().parent.parent / "lib" / "libskyrim_sfno.so"
EXPORTS = ["sksfno_abi_version", "sksfno_prepare_weight", "sksfno_gemm_run", "sksfno_instance_norm",
           "sksfno_chain_dims", "sksfno_prepare_chain_weights", "sksfno_instance_stats", "sksfno_chain_run"]
CHAIN_ENC, CHAIN_MLP, CHAIN_TAIL = 0, 1, 2


class GemmDesc(ctypes.Structure):
    _fields_ = [("a", ctypes.c_void_p), ("a_sb", ctypes.c_longlong), ("a_m1", ctypes.c_int),
                ("a_sm", ctypes.c_longlong), ("a_sm2", ctypes.c_longlong), ("a_sk", ctypes.c_longlong),
                ("w", ctypes.c_void_p), ("w_sb", ctypes.c_longlong), ("w_plane", ctypes.c_longlong), ("ldw", ctypes.c_int),
                ("bias", ctypes.c_void_p), ("res_pre", ctypes.c_void_p), ("res_post", ctypes.c_void_p),
                ("out", ctypes.c_void_p), ("o_sb", ctypes.c_longlong), ("o_m1", ctypes.c_int),
                ("o_sm", ctypes.c_longlong), ("o_sm2", ctypes.c_longlong), ("o_sn", ctypes.c_longlong),
                ("M", ctypes.c_int), ("N", ctypes.c_int), ("K", ctypes.c_int), ("batch", ctypes.c_int), ("act", ctypes.c_int),
                ("k_lo_step", ctypes.c_int), ("m_cap0", ctypes.c_int), ("m_cap_step", ctypes.c_int),
                ("a_kscale", ctypes.c_void_p), ("a_kshift", ctypes.c_void_p),
                ("a2", ctypes.c_void_p), ("a2_sk", ctypes.c_longlong), ("a2_k_split", ctypes.c_int), ("terms", ctypes.c_int)]


class ChainDesc(ctypes.Structure):                      # sksfno_chain
    _fields_ = [("mode", ctypes.c_int), ("shape", ctypes.c_int), ("y", ctypes.c_void_p), ("x", ctypes.c_void_p), ("res", ctypes.c_void_p),
                ("out", ctypes.c_void_p), ("HW", ctypes.c_longlong), ("C", ctypes.c_int), ("KX", ctypes.c_int), ("OUT", ctypes.c_int),
                ("w1f", ctypes.c_void_p), ("w2f", ctypes.c_void_p), ("v1f", ctypes.c_void_p), ("v2f", ctypes.c_void_p), ("tab", ctypes.c_void_p)]


_lib = None


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("SKYRIM_SFNO_LIB", str(_LIB_PATH))
    if not os.path.exists(path):
        raise RuntimeError(f"{path} not found: build the HIP library first (python -c 'import __graft_entry__ as g; g.build()')")
    lib = ctypes.CDLL(path)
    lib.sksfno_abi_version.restype = ctypes.c_int
    lib.sksfno_prepare_weight.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p]
    lib.sksfno_gemm_run.argtypes = [ctypes.POINTER(GemmDesc), ctypes.c_void_p]
    lib.sksfno_instance_norm.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                         ctypes.c_longlong, ctypes.c_float, ctypes.c_void_p]
    lib.sksfno_chain_dims.argtypes = [ctypes.c_int] + [ctypes.POINTER(ctypes.c_int)] * 4
    lib.sksfno_prepare_chain_weights.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                                 ctypes.c_void_p, ctypes.c_void_p]
    lib.sksfno_instance_stats.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int, ctypes.c_longlong, ctypes.c_float, ctypes.c_void_p]
    lib.sksfno_chain_run.argtypes = [ctypes.POINTER(ChainDesc), ctypes.c_void_p]
    for name in EXPORTS:
        getattr(lib, name).restype = ctypes.c_int
    _lib = lib
    return lib


def _check(code: int, what: str):
    if code != 0:
        raise RuntimeError(f"{what} failed with code {code}")


_BIG = 1 << 30       # "no split" value for the two-level row index


class _Weight:
    """A constant matrix [batch][N][K] prepared as fp16 hi/lo planes on the device."""

    def __init__(self, eng, w: torch.Tensor):
        w = w.float().contiguous()
        if w.dim() == 2:
            w = w[None]
        self.batch, self.N, self.K = w.shape
        self.ldw = (self.K + 7) // 8 * 8
        per = self.N * self.ldw
        self.plane = self.batch * per
        self.w_sb = per
        self.buf = torch.empty(2 * self.plane, dtype=torch.float16, device=eng.device)
        chunk = max(1, (256 << 20) // (self.N * self.K * 4))          # upload at most ~256 MB of fp32 at a time
        for b0 in range(0, self.batch, chunk):
            src = w[b0:b0 + chunk].to(eng.device)
            for j in range(src.shape[0]):
                dst = self.buf.data_ptr() + 2 * (b0 + j) * per
                _check(eng.lib.sksfno_prepare_weight(src[j].data_ptr(), self.K, 1, self.N, self.K, dst, self.plane, self.ldw, eng._stream()),
                       "sksfno_prepare_weight")
            torch.cuda.current_stream(eng.device).synchronize()


def chain_shapes(lib) -> list[tuple[int, int, int, int]]:
    """(CP, HP, KXP, OP) of the shape classes the fused pixel-wise chains are compiled for, smallest last."""
    out = []
    for k in range(2):
        v = [ctypes.c_int() for _ in range(4)]
        _check(lib.sksfno_chain_dims(k, *[ctypes.byref(t) for t in v]), "sksfno_chain_dims")
        out.append(tuple(t.value for t in v))
    return out


class _Pair:
    """An expand / contract pair W1 [H][K], W2 [N][H] (zero-padded fp32) as fragment-order fp16 hi/lo planes (sfno_chain.hip)."""

    def __init__(self, eng, w1: torch.Tensor, w2: torch.Tensor):
        H, K = w1.shape
        N = w2.shape[0]
        assert w2.shape[1] == H and K % 32 == 0 and H % 32 == 0 and N % 32 == 0
        a, b = w1.float().contiguous().to(eng.device), w2.float().contiguous().to(eng.device)
        self.w1f = torch.empty(2 * H * K, dtype=torch.float16, device=eng.device)
        self.w2f = torch.empty(2 * N * H, dtype=torch.float16, device=eng.device)
        _check(eng.lib.sksfno_prepare_chain_weights(a.data_ptr(), b.data_ptr(), K, H, N, self.w1f.data_ptr(), self.w2f.data_ptr(), eng._stream()),
               "sksfno_prepare_chain_weights")
        torch.cuda.current_stream(eng.device).synchronize()


def _pad2(m: torch.Tensor, rows: int, cols: int) -> torch.Tensor:
    out = torch.zeros(rows, cols, dtype=torch.float64)
    out[: m.shape[0], : m.shape[1]] = m
    return out


def _pad1(v: torch.Tensor, n: int) -> torch.Tensor:
    out = torch.zeros(n, dtype=torch.float64)
    out[: v.shape[0]] = v
    return out


class SfnoEngine:
    def __init__(self, cfg: SfnoConfig | None = None, device: str | torch.device = "cuda:0", terms: int = 3, fused: bool | None = None):
        """``terms``: MFMA terms per GEMM -- 3: activations and constants as fp16 hi/lo pairs (fp32-class, ~1e-6 vs the oracle);
        2: activations rounded to one fp16 plane (faster; error measured in tests/test_sfno_gpu.py).
        ``fused``: encoder, block MLPs (+ norm1) and decoder as one pixel-wise chain kernel each (sfno_chain.hip) instead of GEMM by
        GEMM; default on (``SKYRIM_SFNO_UNFUSED=1`` switches it off) wherever the widths fit a compiled shape class and the pixel count
        of the grid is a multiple of 16 -- otherwise that stage runs GEMM by GEMM."""
        self.cfg = cfg or SfnoConfig()
        if terms not in (2, 3):
            raise ValueError("terms must be 2 or 3")
        self.terms = terms
        self.fused = (os.environ.get("SKYRIM_SFNO_UNFUSED", "0") != "1") if fused is None else bool(fused)
        self.chain = None                 # shape class of the fused chains, set by load_params
        # layout of the longitude spectrum between the DFT and Legendre GEMMs: "order" = [order, re/im][C][lat], "channel" = [C][order, re/im][lat]
        # grid-changing blocks: the inner skip as a channel mix of the SH coefficients instead of a 1x1 convolution on the output grid
        self.skip_in_spectrum = os.environ.get("SKYRIM_SFNO_GRID_SKIP", "0") != "1"
        self.f_ana = os.environ.get("SKSFNO_F_ANA", "order")
        self.f_syn = os.environ.get("SKSFNO_F_SYN", "order")
        if not torch.cuda.is_available():
            raise RuntimeError("SfnoEngine needs an MI355X: the SFNO path has no CPU fallback")
        self.lib = load_library()
        self.device = torch.device(device)
        self.prepared = False
        self.profiling = False            # per-launch timing with events on the launch stream (bench.py / tools)
        self._events = []
        c = self.cfg
        if c.num_layers < 2:
            raise ValueError("num_layers >= 2 (the first block goes to the internal grid, the last one back)")
        self.state_shape = (c.in_chans, c.n_lat, c.n_lon)
        self._label = "gemm"

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def release(self):
        """Drop every prepared matrix, table and work buffer (GlobalModel.release_model).  The C ABI holds no state of its own -- all device
        memory is torch tensors owned here -- so this IS the teardown; the engine is unusable until ``load_params`` runs again."""
        keep = ("cfg", "terms", "fused", "skip_in_spectrum", "f_ana", "f_syn", "lib", "device", "state_shape", "_label", "profiling")
        kept = {k: v for k, v in vars(self).items() if k in keep}
        self.__dict__.clear()
        self.__dict__.update(kept)
        self.chain = None
        self.prepared = False
        self._events = []

    # ---- prepare ---------------------------------------------------------------------------------- #
    def load_params(self, params: dict):
        c = self.cfg
        for name, shape in param_spec(c):
            if name not in params or tuple(params[name].shape) != tuple(shape):
                raise ValueError(f"parameter {name}: expected shape {shape}, got {tuple(params[name].shape) if name in params else None}")
        p = {k: v.double() for k, v in params.items()}
        dev = self.device
        with torch.cuda.device(dev):
            f32 = lambda t: t.float().contiguous().to(dev)  # noqa: E731
            mean, std = p["norm.mean"], p["norm.std"]
            # the input normalisation is applied by the GEMM's loader BEFORE the fp16 split (raw geopotential / pressure exceed
            # the fp16 range), the output de-normalisation is folded into the decoder's last matrix
            self.in_scale, self.in_shift = f32(1.0 / std), f32(-mean / std)
            self.enc1 = _Weight(self, p["encoder.fc1.weight"])
            self.enc1_b = f32(p["encoder.fc1.bias"])
            self.enc2 = _Weight(self, p["encoder.fc2.weight"])
            self.pos = f32(p["pos_embed"])
            e = c.embed_dim
            self.blocks = []
            for i in range(c.num_layers):
                g = lambda n: p[f"blocks.{i}.{n}"]  # noqa: E731,B023
                fw = g("filter.weight")                               # [in][out][l][2]
                wr, wi = fw[..., 0].permute(2, 1, 0), fw[..., 1].permute(2, 1, 0)      # [l][out][in]
                mix = torch.empty(c.lmax, 2 * e, 2 * e, dtype=torch.float32)
                mix[:, :e, :e] = wr                                   # rows (re/im out, channel out), cols (re/im in, channel in)
                mix[:, :e, e:] = -wi
                mix[:, e:, :e] = wi
                mix[:, e:, e:] = wr
                if self.skip_in_spectrum and i in (0, c.num_layers - 1):
                    # grid-changing block: the inner skip is a channel mix of the coefficients (see step()) -- the same matrix for every
                    # degree and for the real and the imaginary part, i.e. one more term on the diagonal blocks of the dhconv matrices
                    ws = g("inner_skip.weight").float()
                    mix[:, :e, :e] += ws
                    mix[:, e:, e:] += ws
                self.blocks.append(dict(
                    n0_g=f32(g("norm0.weight")), n0_b=f32(g("norm0.bias")), n1_g=f32(g("norm1.weight")), n1_b=f32(g("norm1.bias")),
                    mix=_Weight(self, mix), skip=_Weight(self, g("inner_skip.weight")), skip_b=f32(g("inner_skip.bias")),
                    fc1=_Weight(self, g("mlp.fc1.weight")), fc1_b=f32(g("mlp.fc1.bias")),
                    fc2=_Weight(self, g("mlp.fc2.weight")), fc2_b=f32(g("mlp.fc2.bias")),
                    skip_b00=f32(g("inner_skip.bias") * (4.0 * np.pi) ** 0.5)))        # the bias as the (0, 0) coefficient of a constant field
                del mix
            self.zero_a = torch.zeros(8, dtype=torch.float32, device=dev)
            self.zero_w = _Weight(self, torch.zeros(e, 8))
            # decoder.fc1 acts on concat(features, normalised input): one GEMM with two A sources along K (k < e: features,
            # k >= e: the raw state, normalised by the loader's per-k affine) when e is a multiple of 8, else two GEMMs
            wd = p["decoder.fc1.weight"]
            self.dec_fused = e % 8 == 0
            if self.dec_fused:
                self.dec1 = _Weight(self, wd)
                self.dec_scale = f32(torch.cat([torch.ones(e, dtype=torch.float64), 1.0 / std]))
                self.dec_shift = f32(torch.cat([torch.zeros(e, dtype=torch.float64), -mean / std]))
            else:
                self.dec1a = _Weight(self, wd[:, :e])
                self.dec1b = _Weight(self, wd[:, e:])
            self.dec1_b = f32(p["decoder.fc1.bias"])
            self.dec2 = _Weight(self, p["decoder.fc2.weight"] * std[: c.out_chans, None])
            self.dec2_b = f32(mean[: c.out_chans])
            self._prepare_chains(p, mean, std)
            # transforms: outer (equiangular 721 x 1440) and inner (Legendre-Gauss h x w)
            self.tr = {}
            for key, (nlat, nlon, grid) in {"outer": (c.n_lat, c.n_lon, "equiangular"), "inner": (c.h, c.w, "legendre-gauss")}.items():
                m = ShtMatrices(nlat, nlon, c.lmax, c.mmax, grid)
                self.tr[key] = dict(n_lat=nlat, n_lon=nlon,
                                    dft=_Weight(self, torch.from_numpy(m.dft)), idft=_Weight(self, torch.from_numpy(m.idft)),
                                    ana=_Weight(self, torch.from_numpy(m.analysis)), syn=_Weight(self, torch.from_numpy(m.synthesis)))
            # work buffers
            hw_o, hw_i = c.n_lat * c.n_lon, c.h * c.w
            hid = e * c.mlp_ratio
            buf = lambda n: torch.empty(n, dtype=torch.float32, device=dev)  # noqa: E731
            self.b_y, self.b_xn, self.b_sp, self.b_res = buf(e * hw_o), buf(e * hw_o), buf(e * hw_o), buf(e * hw_o)
            self.b_hid = buf(hid * hw_i)                  # MLP hidden on the internal grid
            self.b_hid_outer = buf(hid * hw_o)            # MLP hidden of the last block (outer grid)
            self.ldl = (c.n_lat + 3) // 4 * 4                 # padded latitude count of the longitude-spectrum buffer
            self.b_f = buf(2 * c.mmax * e * self.ldl)
            self.b_coef = buf(c.lmax * c.mmax * 2 * e)
            self.b_mixed = torch.zeros(c.lmax * c.mmax * 2 * e, dtype=torch.float32, device=dev)     # rows m > l are never written
            self.b_out = buf(c.out_chans * hw_o)
            torch.cuda.current_stream(dev).synchronize()
        self.prepared = True

    def _prepare_chains(self, p: dict, mean: torch.Tensor, std: torch.Tensor):
        """Fragment-order weights and tables of the fused pixel-wise chains (include/skyrim_sfno.h, ABI v2)."""
        c = self.cfg
        e, hid = c.embed_dim, c.embed_dim * c.mlp_ratio
        self.chain = None
        if not self.fused or self.terms != 3:
            return
        for k, (CP, HP, KXP, OP) in reversed(list(enumerate(chain_shapes(self.lib)))):       # smallest class that fits
            if e <= CP and hid <= HP and c.in_chans <= KXP and c.out_chans <= OP:
                self.chain, self.chain_dims = k, (CP, HP, KXP, OP)
                break
        if self.chain is None:
            return
        CP, HP, KXP, OP = self.chain_dims
        dev = self.device
        tab = lambda *parts: torch.cat(parts).float().contiguous().to(dev)  # noqa: E731
        # encoder: K0 = KXP, H0 = CP
        self.ch_enc = _Pair(self, _pad2(p["encoder.fc1.weight"], CP, KXP), _pad2(p["encoder.fc2.weight"], CP, CP))
        self.ch_enc_tab = tab(_pad1(1.0 / std, KXP), _pad1(-mean / std, KXP), _pad1(p["encoder.fc1.bias"], CP), torch.zeros(CP, dtype=torch.float64))
        for i, blk in enumerate(self.blocks):
            g = lambda n: p[f"blocks.{i}.{n}"]  # noqa: E731,B023
            blk["ch_mlp"] = _Pair(self, _pad2(g("mlp.fc1.weight"), HP, CP), _pad2(g("mlp.fc2.weight"), CP, HP))
            parts = [torch.zeros(2 * CP, dtype=torch.float64), _pad1(g("mlp.fc1.bias"), HP), _pad1(g("mlp.fc2.bias"), CP)]    # scale / shift: per step
            if i == c.num_layers - 1:
                wd = p["decoder.fc1.weight"]
                v1 = torch.zeros(CP, CP + KXP, dtype=torch.float64)
                v1[:e, :e] = wd[:, :e]
                v1[:e, CP:CP + c.in_chans] = wd[:, e:]
                blk["ch_dec"] = _Pair(self, v1, _pad2(p["decoder.fc2.weight"] * std[: c.out_chans, None], OP, CP))
                parts += [_pad1(1.0 / std, KXP), _pad1(-mean / std, KXP), _pad1(p["decoder.fc1.bias"], CP), _pad1(mean[: c.out_chans], OP)]
            blk["ch_tab"] = tab(*parts)

    # ---- launches ---------------------------------------------------------------------------------- #
    def _mark(self, label: str, flops: float = 0.0, nbytes: float = 0.0):
        if self.profiling:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream(self.device))
            self._events.append((label, ev, flops, nbytes))

    def profile_read(self) -> list[dict]:
        """Per-stage totals since profiling was switched on: [{name, launches, total_ms, flops, bytes}] (stage = label of the launch)."""
        torch.cuda.synchronize(self.device)
        out: dict[str, dict] = {}
        for (label, ev, fl, by), (_, nxt, _, _) in zip(self._events[:-1], self._events[1:]):
            if label == "end":
                continue
            d = out.setdefault(label, dict(name=label, launches=0, total_ms=0.0, flops=0.0, bytes=0.0))
            d["launches"] += 1
            d["total_ms"] += ev.elapsed_time(nxt)
            d["flops"] += fl
            d["bytes"] += by
        self._events = []
        return list(out.values())

    def _gemm(self, a, W: _Weight, out, M, K, N, *, a_sm, a_sk, o_sm, o_sn, batch=1, a_sb=0, o_sb=0, a_m1=_BIG, a_sm2=0,
              o_m1=_BIG, o_sm2=0, bias=None, res_pre=None, res_post=None, act=0, a_off=0, o_off=0, w_batched=None, k_lo_step=0, m_cap0=0, m_cap_step=0, a_kscale=None, a_kshift=None, a2=None, a2_sk=0, a2_k_split=0):
        if N != W.N or K != W.K:
            raise ValueError(f"GEMM {M}x{N}x{K} against a prepared [{W.N}][{W.K}] matrix")
        self._mark(self._label, 2.0 * M * N * K * batch, 4.0 * batch * (M * K + M * N * (1 + (res_pre is not None) + (res_post is not None))))
        geom = [a_off, a_sb, a_m1, a_sm, a_sm2, a_sk, (W.w_sb if (batch > 1 if w_batched is None else w_batched) else 0), W.plane, W.ldw,
                o_off, o_sb, o_m1, o_sm, o_sm2, o_sn, M, N, K, batch, act, k_lo_step, m_cap0, m_cap_step, a2_sk, a2_k_split, self.terms]
        ops.hip.sfno_gemm(a, W.buf, out, bias, res_pre, res_post, a_kscale, a_kshift, a2, geom)

    def _norm(self, x, g, b, out, C, HW):
        self._mark("norm", 8.0 * C * HW, 16.0 * C * HW)
        ops.hip.sfno_instance_norm(x, g, b, out, C, HW, self.cfg.eps)

    def _chain(self, mode, label, y, res, out, hw, pair, tab, x=None, dec=None):
        """One fused pixel-wise chain over hw pixels (sksfno_chain_run)."""
        c = self.cfg
        e, hid = c.embed_dim, c.embed_dim * c.mlp_ratio
        macs = {CHAIN_ENC: c.in_chans * e + e * e, CHAIN_MLP: 2 * e * hid, CHAIN_TAIL: 2 * e * hid + (e + c.in_chans) * e + e * c.out_chans}[mode]
        chans = {CHAIN_ENC: c.in_chans + 2 * e, CHAIN_MLP: 3 * e, CHAIN_TAIL: 2 * e + c.in_chans + c.out_chans}[mode]
        self._mark(label, 2.0 * hw * macs, 4.0 * hw * chans)
        ops.hip.sfno_chain(mode, self.chain, y, x, res, out, hw, e, c.in_chans, c.out_chans, pair.w1f, pair.w2f,
                           None if dec is None else dec.w1f, None if dec is None else dec.w2f, tab)

    def _stats(self, x, g, b, tab, C, HW):
        """Instance-norm statistics of x as the consumer's per-channel affine: scale -> tab[0:C], shift -> tab[CP:CP + C]."""
        self._mark("norm", 4.0 * C * HW, 4.0 * C * HW)
        ops.hip.sfno_instance_stats(x, g, b, tab, self.chain_dims[0], C, HW, self.cfg.eps)

    def _pointwise(self, a, W, out, hw, cin, cout, label="conv1x1", **kw):
        """1x1 convolution on [C][hw] activations: rows = pixels (contiguous), k = channel (stride hw)."""
        self._label = label
        self._gemm(a, W, out, hw, cin, cout, a_sm=1, a_sk=hw, o_sm=1, o_sn=hw, **kw)

    def _analysis(self, x, tr, C):
        """[C][H][W] -> SH coefficients [l][m][re/im][C] in self.b_coef."""
        c = self.cfg
        H, Wd, Mm, L, ldl = tr["n_lat"], tr["n_lon"], c.mmax, c.lmax, self.ldl
        # truncated DFT, one batch per channel: rows = latitudes, k = longitude -> spectrum [order, re/im][C][ldl] (order-major, so
        # that the Legendre GEMM of one order reads k = latitude contiguously)
        self._label = "dft"
        if self.f_ana == "channel":        # spectrum [C][order, re/im][ldl]: the DFT of a channel writes one dense region
            self._gemm(x, tr["dft"], self.b_f, H, Wd, 2 * Mm, batch=C, a_sb=H * Wd, a_sm=Wd, a_sk=1, o_sb=2 * Mm * ldl, o_sm=1, o_sn=ldl, w_batched=False)
            self._label = "legendre_analysis"
            self._gemm(self.b_f, tr["ana"], self.b_coef, 2 * C, H, L, batch=Mm, a_sb=2 * ldl, a_m1=C, a_sm=2 * Mm * ldl, a_sm2=ldl, a_sk=1,
                       o_sb=2 * C, o_sm=1, o_sn=Mm * 2 * C)
            return
        self._gemm(x, tr["dft"], self.b_f, H, Wd, 2 * Mm, batch=C, a_sb=H * Wd, a_sm=Wd, a_sk=1, o_sb=ldl, o_sm=1, o_sn=C * ldl, w_batched=False)
        # per order m: rows (re/im, channel), k = latitude
        self._label = "legendre_analysis"
        self._gemm(self.b_f, tr["ana"], self.b_coef, 2 * C, H, L, batch=Mm, a_sb=2 * C * ldl, a_sm=ldl, a_sk=1,
                   o_sb=2 * C, o_sm=1, o_sn=Mm * 2 * C)

    def _synthesis(self, coef, tr, out, C, **kw):
        """SH coefficients [l][m][re/im][C] -> [C][H][W] (+ epilogue options of the last GEMM)."""
        c = self.cfg
        H, Wd, Mm, L, ldl = tr["n_lat"], tr["n_lon"], c.mmax, c.lmax, self.ldl
        self._label = "legendre_synthesis"
        # order m only has degrees l >= m: the contraction over l starts at (the 32-aligned floor of) m
        if self.f_syn == "channel":        # spectrum [C][order, re/im][ldl]: the inverse DFT of a channel reads one dense region
            self._gemm(coef, tr["syn"], self.b_f, 2 * C, L, H, batch=Mm, a_sb=2 * C, a_sm=1, a_sk=Mm * 2 * C,
                       o_sb=2 * ldl, o_m1=C, o_sm=2 * Mm * ldl, o_sm2=ldl, o_sn=1, k_lo_step=1)
            self._label = "idft"
            self._gemm(self.b_f, tr["idft"], out, H, 2 * Mm, Wd, batch=C, a_sb=2 * Mm * ldl, a_sm=1, a_sk=ldl, o_sb=H * Wd, o_sm=Wd, o_sn=1, w_batched=False, **kw)
            return
        self._gemm(coef, tr["syn"], self.b_f, 2 * C, L, H, batch=Mm, a_sb=2 * C, a_sm=1, a_sk=Mm * 2 * C,
                   o_sb=2 * C * ldl, o_sm=ldl, o_sn=1, k_lo_step=1)
        self._label = "idft"
        self._gemm(self.b_f, tr["idft"], out, H, 2 * Mm, Wd, batch=C, a_sb=ldl, a_sm=1, a_sk=C * ldl, o_sb=H * Wd, o_sm=Wd, o_sn=1, w_batched=False, **kw)

    def step(self, x: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        """One 6-h step: fp32 (in_chans, n_lat, n_lon) on the engine device -> (out_chans, n_lat, n_lon)."""
        if not self.prepared:
            raise RuntimeError("SfnoEngine.step before load_params: not prepared")
        c = self.cfg
        if x.device != self.device or x.dtype != torch.float32 or tuple(x.shape) != self.state_shape or not x.is_contiguous():
            raise ValueError(f"expected a contiguous float32 tensor of shape {self.state_shape} on {self.device}")
        e, hid = c.embed_dim, c.embed_dim * c.mlp_ratio
        hw_o = c.n_lat * c.n_lon
        with torch.cuda.device(self.device):
            y = out if out is not None else torch.empty((c.out_chans, c.n_lat, c.n_lon), dtype=torch.float32, device=self.device)
            if y.device != self.device or y.dtype != torch.float32 or tuple(y.shape) != (c.out_chans, c.n_lat, c.n_lon) or not y.is_contiguous():
                raise ValueError("bad output tensor")
            fuse_o = self.chain is not None and hw_o % 16 == 0          # outer-grid chains
            # encoder: GELU(W1' x + b1') -> W2 . + position embedding
            if fuse_o:
                self._chain(CHAIN_ENC, "encoder", x, self.pos, self.b_y, hw_o, self.ch_enc, self.ch_enc_tab)
            else:
                self._pointwise(x, self.enc1, self.b_sp, hw_o, c.in_chans, e, label="encoder", bias=self.enc1_b, act=1,
                                a_kscale=self.in_scale, a_kshift=self.in_shift)
                self._pointwise(self.b_sp, self.enc2, self.b_y, hw_o, e, e, label="encoder", res_post=self.pos)
            cur = self.b_y
            done = False
            for i, blk in enumerate(self.blocks):
                tin = self.tr["outer"] if i == 0 else self.tr["inner"]
                tout = self.tr["outer"] if i == c.num_layers - 1 else self.tr["inner"]
                hw_in, hw_out = tin["n_lat"] * tin["n_lon"], tout["n_lat"] * tout["n_lon"]
                self._norm(cur, blk["n0_g"], blk["n0_b"], self.b_xn, e, hw_in)
                self._analysis(self.b_xn, tin, e)
                if tin is tout:
                    res = self.b_xn
                else:                                               # the residual is the normalised input on the OUTPUT grid
                    self._synthesis(self.b_coef, tout, self.b_res, e)
                    res = self.b_res
                # dhconv: per degree l, rows = orders m, k = (in channel, re/im) -> (out channel, re/im)
                self._label = "dhconv"
                self._gemm(self.b_coef, blk["mix"], self.b_mixed, c.mmax, 2 * e, 2 * e, batch=c.lmax, a_sb=c.mmax * 2 * e, a_sm=2 * e, a_sk=1,
                           o_sb=c.mmax * 2 * e, o_sm=2 * e, o_sn=1, m_cap0=1, m_cap_step=1)      # degree l has orders m <= l only
                outer = "_outer" if tout is self.tr["outer"] else ""
                if tin is not tout and self.skip_in_spectrum:
                    # the residual of a grid-changing block is iSHT(coef): band-limited, so the 1x1 inner skip commutes with the synthesis --
                    # skip(iSHT(coef)) = iSHT(W_skip coef), a channel mix of the COEFFICIENTS with the same matrix for every degree: it is
                    # part of this block's dhconv matrices (load_params).  What is left: the bias on the (l, m) = (0, 0) coefficient (a
                    # constant field b is b * sqrt(4 pi) there), and the GELU in the inverse DFT's epilogue
                    self._label = "inner_skip" + outer
                    self._gemm(self.zero_a, self.zero_w, self.b_mixed, 1, 8, e, a_sm=8, a_sk=1, o_sm=e, o_sn=1, bias=blk["skip_b00"], res_post=self.b_mixed)
                    self._synthesis(self.b_mixed, tout, self.b_y, e, act=1)
                else:
                    self._synthesis(self.b_mixed, tout, self.b_sp, e)
                    # GELU(filter output + inner skip(residual))
                    self._pointwise(res, blk["skip"], self.b_y, hw_out, e, e, label="inner_skip" + outer, bias=blk["skip_b"], res_pre=self.b_sp, act=1)
                if self.chain is not None and hw_out % 16 == 0:
                    # norm1 as the chain's input affine; the last block's chain runs on into the decoder and writes the next state
                    self._stats(self.b_y, blk["n1_g"], blk["n1_b"], blk["ch_tab"], e, hw_out)
                    if i == c.num_layers - 1:
                        self._chain(CHAIN_TAIL, "mlp_decoder", self.b_y, res, y, hw_out, blk["ch_mlp"], blk["ch_tab"], x=x, dec=blk["ch_dec"])
                        done = True
                    else:
                        self._chain(CHAIN_MLP, "mlp" + outer, self.b_y, res, self.b_y, hw_out, blk["ch_mlp"], blk["ch_tab"])
                else:
                    self._norm(self.b_y, blk["n1_g"], blk["n1_b"], self.b_sp, e, hw_out)
                    hbuf = self.b_hid_outer if tout is self.tr["outer"] else self.b_hid
                    self._pointwise(self.b_sp, blk["fc1"], hbuf, hw_out, e, hid, label="mlp" + outer, bias=blk["fc1_b"], act=1)
                    self._pointwise(hbuf, blk["fc2"], self.b_y, hw_out, hid, e, label="mlp" + outer, bias=blk["fc2_b"], res_post=res)
                cur = self.b_y
            if done:
                self._mark("end")
                return y
            # decoder on concat(cur, normalised input): W_a cur + b' , then GELU(W_b' x + .), then W2' . + mean
            if self.dec_fused:
                self._pointwise(cur, self.dec1, self.b_xn, hw_o, e + c.in_chans, e, label="decoder", bias=self.dec1_b, act=1,
                                a_kscale=self.dec_scale, a_kshift=self.dec_shift, a2=x, a2_sk=hw_o, a2_k_split=e)
            else:
                self._pointwise(cur, self.dec1a, self.b_sp, hw_o, e, e, label="decoder", bias=self.dec1_b)
                self._pointwise(x, self.dec1b, self.b_xn, hw_o, c.in_chans, e, label="decoder", res_pre=self.b_sp, act=1,
                                a_kscale=self.in_scale, a_kshift=self.in_shift)
            target = self.b_out if y.data_ptr() == x.data_ptr() else y
            self._pointwise(self.b_xn, self.dec2, target, hw_o, e, c.out_chans, label="decoder", bias=self.dec2_b)
            if target is self.b_out:
                y.view(-1).copy_(self.b_out)
            self._mark("end")
        return y

    def launches_per_step(self) -> int:
        c = self.cfg
        fo = self.chain is not None and (c.n_lat * c.n_lon) % 16 == 0
        fi = self.chain is not None and (c.h * c.w) % 16 == 0
        n = 1 if fo else 2                                    # encoder
        for i in range(c.num_layers):
            last = i == c.num_layers - 1
            change = i in (0, c.num_layers - 1)                   # grid-changing blocks: + residual synthesis (2)
            n += 1 + 2 + 1 + 2 + 1 + (2 if change else 0)       # norm0, analysis, dhconv, synthesis, skip (grid space) or its bias row (spectrum)
            n += 2 if (fo if last else fi) else 3             # statistics + chain  |  norm1 + fc1 + fc2
        if not fo:
            n += 2 if self.cfg.embed_dim % 8 == 0 else 3      # decoder (part of the last chain otherwise)
        return n
