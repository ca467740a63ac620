"""``torch.ops.skyrim_hip.*`` -- the Python-side call path into the HIP kernels ("wrapped as torch.library custom ops").

Every op is a thin dispatcher entry over one ``extern "C"`` launcher of include/skyrim_{pangu,sfno,graphcast}.h: the op validates
its tensors (device, dtype, contiguity), takes the device guard and torch's CURRENT stream of that device, and calls the C ABI with
raw pointers.  Only the CUDA (= ROCm) dispatch key is registered: calling an op with CPU tensors raises NotImplementedError from the
dispatcher -- there is no CPU fallback.  Outputs are written in place into caller-owned tensors (schema ``Tensor(a!)``), so the ops
are stream-ordered, allocation-free and capturable in a HIP graph.

    pangu_step / pangu_patch_embed / pangu_block / pangu_downsample / pangu_upsample / pangu_patch_recover     (ctx = skpangu_ctx*)
    sfno_gemm / sfno_instance_norm / sfno_chain / sfno_instance_stats
    gc_gather_gemm / gc_linear_layer_norm / gc_sum_linear_layer_norm / gc_layer_norm / gc_segment_sum
    gc_edge_update / gc_segment_fixup / gc_node_mlp      (the fused interaction-network updates, csrc/graphcast_fused.hip)
"""
from __future__ import annotations

import ctypes

import torch

_NS = "skyrim_hip"
_lib = torch.library.Library(_NS, "DEF")
_registered = False


def _stream(t: torch.Tensor):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _f32(t: torch.Tensor, what: str, dev=None):
    if t.dtype != torch.float32 or not t.is_contiguous() or (dev is not None and t.device != dev):
        raise ValueError(f"{what}: expected a contiguous float32 tensor on {dev or 'the GPU'}")
    return ctypes.c_void_p(t.data_ptr())


def _opt(t, off_bytes: int = 0):
    return None if t is None else ctypes.c_void_p(t.data_ptr() + off_bytes)


def _ok(code: int, what: str, strerror=None):
    if code != 0:
        msg = strerror(code).decode() if strerror is not None else ""
        raise RuntimeError(f"{what} failed: {msg} (code {code})")


# ---- Pangu ---------------------------------------------------------------------------------------------------------------- #
def _pangu():
    from .pangu import engine
    return engine.load_library()


def _pangu_step(ctx: int, x: torch.Tensor, out: torch.Tensor) -> None:
    lib = _pangu()
    with torch.cuda.device(x.device):
        _ok(lib.skpangu_step(ctypes.c_void_p(ctx), _f32(x, "x"), _f32(out, "out", x.device), _stream(x)), "skpangu_step", lib.skpangu_error_string)


def _pangu_patch_embed(ctx: int, x: torch.Tensor, out: torch.Tensor) -> None:
    lib = _pangu()
    with torch.cuda.device(x.device):
        _ok(lib.skpangu_patch_embed(ctypes.c_void_p(ctx), _f32(x, "x"), _f32(out, "out", x.device), _stream(x)), "skpangu_patch_embed", lib.skpangu_error_string)


def _pangu_block(ctx: int, layer: int, block: int, x: torch.Tensor) -> None:
    lib = _pangu()
    with torch.cuda.device(x.device):
        _ok(lib.skpangu_block(ctypes.c_void_p(ctx), layer, block, _f32(x, "x"), _stream(x)), "skpangu_block", lib.skpangu_error_string)


def _pangu_downsample(ctx: int, x1: torch.Tensor, out: torch.Tensor) -> None:
    lib = _pangu()
    with torch.cuda.device(x1.device):
        _ok(lib.skpangu_downsample(ctypes.c_void_p(ctx), _f32(x1, "x1"), _f32(out, "out", x1.device), _stream(x1)), "skpangu_downsample", lib.skpangu_error_string)


def _pangu_upsample(ctx: int, x2: torch.Tensor, out: torch.Tensor) -> None:
    lib = _pangu()
    with torch.cuda.device(x2.device):
        _ok(lib.skpangu_upsample(ctypes.c_void_p(ctx), _f32(x2, "x2"), _f32(out, "out", x2.device), _stream(x2)), "skpangu_upsample", lib.skpangu_error_string)


def _pangu_patch_recover(ctx: int, skip: torch.Tensor, x4: torch.Tensor, out: torch.Tensor) -> None:
    lib = _pangu()
    with torch.cuda.device(skip.device):
        _ok(lib.skpangu_patch_recover(ctypes.c_void_p(ctx), _f32(skip, "skip"), _f32(x4, "x4", skip.device), _f32(out, "out", skip.device), _stream(skip)),
            "skpangu_patch_recover", lib.skpangu_error_string)


# ---- SFNO ------------------------------------------------------------------------------------------------------------------ #
# geometry vector of sfno_gemm, in this order
SFNO_GEMM_GEOM = ("a_off", "a_sb", "a_m1", "a_sm", "a_sm2", "a_sk", "w_sb", "w_plane", "ldw", "o_off", "o_sb", "o_m1", "o_sm", "o_sm2", "o_sn",
                  "M", "N", "K", "batch", "act", "k_lo_step", "m_cap0", "m_cap_step", "a2_sk", "a2_k_split", "terms")


def _sfno_gemm(a, w, out, bias, res_pre, res_post, a_kscale, a_kshift, a2, geom) -> None:
    from .sfno import engine
    lib = engine.load_library()
    if len(geom) != len(SFNO_GEMM_GEOM):
        raise ValueError(f"sfno_gemm: geom has {len(geom)} entries, expected {len(SFNO_GEMM_GEOM)}")
    g = dict(zip(SFNO_GEMM_GEOM, geom))
    for t, what in ((a, "a"), (out, "out"), (bias, "bias"), (res_pre, "res_pre"), (res_post, "res_post"), (a_kscale, "a_kscale"), (a_kshift, "a_kshift"), (a2, "a2")):
        if t is not None:
            _f32(t, what, a.device)
    if w.dtype != torch.float16 or w.device != a.device:
        raise ValueError("sfno_gemm: w must be the fp16 hi/lo planes of sksfno_prepare_weight on the same device")
    d = engine.GemmDesc(_opt(a, 4 * g["a_off"]), g["a_sb"], g["a_m1"], g["a_sm"], g["a_sm2"], g["a_sk"],
                        w.data_ptr(), g["w_sb"], g["w_plane"], g["ldw"], _opt(bias), _opt(res_pre, 4 * g["o_off"]), _opt(res_post, 4 * g["o_off"]),
                        _opt(out, 4 * g["o_off"]), g["o_sb"], g["o_m1"], g["o_sm"], g["o_sm2"], g["o_sn"], g["M"], g["N"], g["K"], g["batch"], g["act"],
                        g["k_lo_step"], g["m_cap0"], g["m_cap_step"], _opt(a_kscale), _opt(a_kshift), _opt(a2), g["a2_sk"], g["a2_k_split"], g["terms"])
    with torch.cuda.device(a.device):
        _ok(lib.sksfno_gemm_run(ctypes.byref(d), _stream(a)), "sksfno_gemm_run")


def _sfno_instance_norm(x, gamma, beta, out, C: int, HW: int, eps: float) -> None:
    from .sfno import engine
    lib = engine.load_library()
    with torch.cuda.device(x.device):
        _ok(lib.sksfno_instance_norm(_f32(x, "x"), _f32(gamma, "gamma", x.device), _f32(beta, "beta", x.device), _f32(out, "out", x.device), C, HW, eps, _stream(x)),
            "sksfno_instance_norm")


def _sfno_chain(mode: int, shape: int, y, x, res, out, HW: int, C: int, KX: int, OUT: int, w1f, w2f, v1f, v2f, tab) -> None:
    from .sfno import engine
    lib = engine.load_library()
    dev = y.device
    for t, what in ((y, "y"), (x, "x"), (res, "res"), (out, "out"), (tab, "tab")):
        if t is not None:
            _f32(t, what, dev)
    for t, what in ((w1f, "w1f"), (w2f, "w2f"), (v1f, "v1f"), (v2f, "v2f")):
        if t is not None and (t.dtype != torch.float16 or t.device != dev or not t.is_contiguous()):
            raise ValueError(f"sfno_chain: {what} must be the fp16 planes of sksfno_prepare_chain_weights on the same device")
    nin = KX if mode == engine.CHAIN_ENC else C
    nout = OUT if mode == engine.CHAIN_TAIL else C
    if y.numel() < nin * HW or res.numel() < C * HW or out.numel() < nout * HW or (x is not None and x.numel() < KX * HW):
        raise ValueError("sfno_chain: an activation tensor is smaller than channels x HW")
    d = engine.ChainDesc(mode, shape, y.data_ptr(), _opt(x), res.data_ptr(), out.data_ptr(), HW, C, KX, OUT, w1f.data_ptr(), w2f.data_ptr(),
                         _opt(v1f), _opt(v2f), tab.data_ptr())
    with torch.cuda.device(dev):
        _ok(lib.sksfno_chain_run(ctypes.byref(d), _stream(y)), "sksfno_chain_run")


def _sfno_instance_stats(x, gamma, beta, tab, shift_off: int, C: int, HW: int, eps: float) -> None:
    from .sfno import engine
    lib = engine.load_library()
    dev = x.device
    if tab.numel() < shift_off + C or x.numel() < C * HW:
        raise ValueError("sfno_instance_stats: tab or x too small")
    with torch.cuda.device(dev):
        _ok(lib.sksfno_instance_stats(_f32(x, "x"), _f32(gamma, "gamma", dev), _f32(beta, "beta", dev), _f32(tab, "tab", dev),
                                      ctypes.c_void_p(tab.data_ptr() + 4 * shift_off), C, HW, eps, _stream(x)), "sksfno_instance_stats")


# ---- GraphCast -------------------------------------------------------------------------------------------------------------- #
def _gc_gather_gemm(src, idx, width, w, w_plane: int, ldw: int, bias, out, M: int, N: int, act: int, kscale, kshift) -> None:
    from .graphcast import engine
    lib = engine.load_library()
    if not (1 <= len(src) <= 3) or len(idx) != len(src) or len(width) != len(src):
        raise ValueError("gc_gather_gemm: 1..3 sources with one (optional) index tensor and one width each")
    d = engine.GatherDesc()
    dev = out.device
    for s, (t, ix, wd) in enumerate(zip(src, idx, width)):
        _f32(t, f"src[{s}]", dev)
        if ix is not None and (ix.dtype != torch.int32 or ix.device != dev or not ix.is_contiguous()):
            raise ValueError("gc_gather_gemm: index tensors are contiguous int32 on the same device")
        d.src[s], d.idx[s] = t.data_ptr(), (ix.data_ptr() if ix is not None else None)
        d.ld[s] = t.shape[-1] if t.dim() == 2 else wd
        d.width[s] = wd
    d.n_src = len(src)
    d.kscale, d.kshift = (kscale.data_ptr() if kscale is not None else None), (kshift.data_ptr() if kshift is not None else None)
    d.w, d.w_plane, d.ldw = w.data_ptr(), w_plane, ldw
    d.bias = _f32(bias, "bias", dev).value
    d.out, d.ldo, d.M, d.N, d.act = _f32(out, "out").value, N, M, N, act
    with torch.cuda.device(dev):
        _ok(lib.skgc_gather_gemm(ctypes.byref(d), _stream(out)), "skgc_gather_gemm")


def _gc_linear_layer_norm(a, lda: int, K: int, w, w_plane: int, ldw: int, bias, gamma, beta, res, out, rows: int) -> None:
    from .graphcast import engine
    lib = engine.load_library()
    dev = out.device
    with torch.cuda.device(dev):
        _ok(lib.skgc_linear_layer_norm(_f32(a, "a", dev), lda, K, ctypes.c_void_p(w.data_ptr()), w_plane, ldw, _f32(bias, "bias", dev), _f32(gamma, "gamma", dev),
                                       _f32(beta, "beta", dev), _opt(res), _f32(out, "out"), rows, _stream(out)), "skgc_linear_layer_norm")


def _gc_sum_linear_layer_norm(src, src_off, ld, idx, K: int, act: int, w, w_plane: int, ldw: int, bias, gamma, beta, res, out, rows: int, group: int = 0) -> None:
    from .graphcast import engine
    lib = engine.load_library()
    dev = out.device
    if not (1 <= len(src) <= 3) or not (len(src) == len(src_off) == len(ld) == len(idx)):
        raise ValueError("gc_sum_linear_layer_norm: 1..3 sources with an element offset, a leading dimension and an (optional) index tensor each")
    d = engine.SumDesc()
    for s, (t, off, l, ix) in enumerate(zip(src, src_off, ld, idx)):
        _f32(t, f"src[{s}]", dev)
        if ix is not None and (ix.dtype != torch.int32 or ix.device != dev or not ix.is_contiguous()):
            raise ValueError("gc_sum_linear_layer_norm: index tensors are contiguous int32 on the same device")
        d.src[s], d.idx[s], d.ld[s] = t.data_ptr() + 4 * off, (ix.data_ptr() if ix is not None else None), l
    d.n_src, d.K, d.act = len(src), K, act
    d.w, d.w_plane, d.ldw = w.data_ptr(), w_plane, ldw
    d.bias = bias.data_ptr() if bias is not None else None
    d.gamma, d.beta = _f32(gamma, "gamma", dev).value, _f32(beta, "beta", dev).value
    d.res = res.data_ptr() if res is not None else None
    d.out, d.rows, d.group = _f32(out, "out").value, rows, group
    with torch.cuda.device(dev):
        _ok(lib.skgc_sum_linear_layer_norm(ctypes.byref(d), _stream(out)), "skgc_sum_linear_layer_norm")


def _gc_layer_norm(x, gamma, beta, res, out, rows: int, N: int) -> None:
    from .graphcast import engine
    lib = engine.load_library()
    dev = out.device
    with torch.cuda.device(dev):
        _ok(lib.skgc_layer_norm(_f32(x, "x", dev), _f32(gamma, "gamma", dev), _f32(beta, "beta", dev), _opt(res), _f32(out, "out"), rows, N, _stream(out)), "skgc_layer_norm")


def _gc_segment_sum(e, offsets, out, acc, n_nodes: int, N: int) -> None:
    from .graphcast import engine
    lib = engine.load_library()
    dev = out.device
    if offsets.dtype != torch.int32 or offsets.device != dev:
        raise ValueError("gc_segment_sum: offsets are int32 on the same device")
    with torch.cuda.device(dev):
        _ok(lib.skgc_segment_sum(_f32(e, "e", dev), ctypes.c_void_p(offsets.data_ptr()), _f32(out, "out"), _opt(acc), n_nodes, N, _stream(out)), "skgc_segment_sum")


def _i32(t, what: str, dev):
    if t.dtype != torch.int32 or t.device != dev or not t.is_contiguous():
        raise ValueError(f"{what}: expected a contiguous int32 tensor on {dev}")
    return t.data_ptr()


def _f16(t, what: str, dev):
    if t.dtype != torch.float16 or t.device != dev or not t.is_contiguous():
        raise ValueError(f"{what}: expected a contiguous float16 tensor on {dev}")
    return t.data_ptr()


def _gc_edge_update(e_in, e_out, term, term_off, ld, idx, recv, w1f, w2f, b2, gamma, beta, agg, heads, rows: int, probe=None, w1_planes: int = 2) -> None:
    from .graphcast import engine
    lib = engine.load_library()
    dev = agg.device
    if len(term) > 2 or not (len(term) == len(term_off) == len(ld) == len(idx)):
        raise ValueError("gc_edge_update: at most two gathered terms, each with an element offset, a leading dimension and an index tensor")
    d = engine.EdgeDesc()
    d.e_in = _f16(e_in, "e_in", dev)
    d.e_out = _f16(e_out, "e_out", dev) if e_out is not None else None
    for s, (t, off, l, ix) in enumerate(zip(term, term_off, ld, idx)):
        _f32(t, f"term[{s}]", dev)
        d.term[s], d.idx[s], d.ld[s] = t.data_ptr() + 4 * off, _i32(ix, f"idx[{s}]", dev), l
        if ix.numel() < rows:
            raise ValueError("gc_edge_update: index tensors hold one entry per packed row")
    d.n_term = len(term)
    if recv.numel() < rows or e_in.numel() < rows * 512 or (e_out is not None and e_out.numel() < rows * 512):
        raise ValueError("gc_edge_update: recv / e_in / e_out are smaller than `rows` packed rows")
    d.recv = _i32(recv, "recv", dev)
    d.w1f = _f16(w1f, "w1f", dev) if w1f is not None else None
    d.w2f = _f16(w2f, "w2f", dev)
    d.b2, d.gamma, d.beta = _f32(b2, "b2", dev).value, _f32(gamma, "gamma", dev).value, _f32(beta, "beta", dev).value
    d.agg = _f32(agg, "agg").value
    if heads is not None:
        if heads.numel() < rows // 128 * 512:
            raise ValueError("gc_edge_update: heads holds one 512-wide row per 128-row tile")
    elif rows > 128 and bool(((recv[127:rows - 1:128] == recv[128:rows:128]) & (recv[128:rows:128] >= 0)).any().item()):
        # ad-hoc callers only (one blocking device read): GraphcastEngine.pack() proves the same thing on the host (fused.continuation_list,
        # padding rows -1 excluded there as here) and always hands over a heads buffer, so the engine's step never comes this way
        raise ValueError("gc_edge_update: a receiver's run continues across a tile boundary -- pass a heads buffer (and run gc_segment_fixup)")
    if agg.numel() < 512 or any(t.numel() - off < l for t, off, l in zip(term, term_off, ld)):
        raise ValueError("gc_edge_update: agg / term buffers are smaller than one row")
    d.heads = _f32(heads, "heads", dev).value if heads is not None else None
    d.rows, d.has_fc1, d.w1_planes = rows, int(w1f is not None), w1_planes
    if w1f is not None and w1f.numel() != w1_planes * 512 * 512:
        raise ValueError("gc_edge_update: w1f holds w1_planes planes of [512][512] in fragment order")
    d.probe = probe.data_ptr() if probe is not None else None
    with torch.cuda.device(dev):
        _ok(lib.skgc_edge_update(ctypes.byref(d), _stream(agg)), "skgc_edge_update")


def _gc_segment_fixup(agg, heads, nodes, first, tiles) -> None:
    from .graphcast import engine
    lib = engine.load_library()
    dev = agg.device
    with torch.cuda.device(dev):
        _ok(lib.skgc_segment_fixup(_f32(agg, "agg"), _f32(heads, "heads", dev), ctypes.c_void_p(_i32(nodes, "nodes", dev)), ctypes.c_void_p(_i32(first, "first", dev)),
                                   ctypes.c_void_p(_i32(tiles, "tiles", dev)), nodes.numel(), _stream(agg)), "skgc_segment_fixup")


def _gc_node_mlp(src, src_off, ld, w1f, w2f, b1, b2, gamma, beta, res, res_off: int, ld_res: int, out, out_off: int, ld_out: int, rows: int) -> None:
    from .graphcast import engine
    lib = engine.load_library()
    dev = out.device
    if not (1 <= len(src) <= 2) or not (len(src) == len(src_off) == len(ld)):
        raise ValueError("gc_node_mlp: one or two fp32 sources, each with an element offset and a leading dimension")
    d = engine.NodeDesc()
    for s, (t, off, l) in enumerate(zip(src, src_off, ld)):
        _f32(t, f"src[{s}]", dev)
        if t.numel() < off + (rows - 1) * l + 512:
            raise ValueError("gc_node_mlp: source smaller than rows x 512")
        d.src[s], d.ld[s] = t.data_ptr() + 4 * off, l
    d.n_src = len(src)
    d.w1f, d.w2f = _f16(w1f, "w1f", dev), _f16(w2f, "w2f", dev)
    if w1f.numel() != 2 * 512 * 512 * len(src) or w2f.numel() != 2 * 512 * 512:
        raise ValueError("gc_node_mlp: weight fragments are [512][512 n_src] and [512][512] with hi/lo planes")
    d.b1, d.b2 = _f32(b1, "b1", dev).value, _f32(b2, "b2", dev).value
    d.gamma, d.beta = _f32(gamma, "gamma", dev).value, _f32(beta, "beta", dev).value
    d.res = (_f32(res, "res", dev).value + 4 * res_off) if res is not None else None
    d.ld_res = ld_res
    if out.numel() < out_off + (rows - 1) * ld_out + 512:
        raise ValueError("gc_node_mlp: out smaller than rows x 512")
    d.out, d.ld_out, d.rows = _f32(out, "out").value + 4 * out_off, ld_out, rows
    with torch.cuda.device(dev):
        _ok(lib.skgc_node_mlp(ctypes.byref(d), _stream(out)), "skgc_node_mlp")


_SCHEMAS = [
    ("pangu_step(int ctx, Tensor x, Tensor(a!) out) -> ()", _pangu_step),
    ("pangu_patch_embed(int ctx, Tensor x, Tensor(a!) out) -> ()", _pangu_patch_embed),
    ("pangu_block(int ctx, int layer, int block, Tensor(a!) x) -> ()", _pangu_block),
    ("pangu_downsample(int ctx, Tensor x1, Tensor(a!) out) -> ()", _pangu_downsample),
    ("pangu_upsample(int ctx, Tensor x2, Tensor(a!) out) -> ()", _pangu_upsample),
    ("pangu_patch_recover(int ctx, Tensor skip, Tensor x4, Tensor(a!) out) -> ()", _pangu_patch_recover),
    ("sfno_gemm(Tensor a, Tensor w, Tensor(a!) out, Tensor? bias, Tensor? res_pre, Tensor? res_post, Tensor? a_kscale, Tensor? a_kshift, Tensor? a2, int[] geom) -> ()",
     _sfno_gemm),
    ("sfno_instance_norm(Tensor x, Tensor gamma, Tensor beta, Tensor(a!) out, int C, int HW, float eps) -> ()", _sfno_instance_norm),
    ("sfno_chain(int mode, int shape, Tensor y, Tensor? x, Tensor res, Tensor(a!) out, int HW, int C, int KX, int OUT, Tensor w1f, Tensor w2f, "
     "Tensor? v1f, Tensor? v2f, Tensor tab) -> ()", _sfno_chain),
    ("sfno_instance_stats(Tensor x, Tensor gamma, Tensor beta, Tensor(a!) tab, int shift_off, int C, int HW, float eps) -> ()", _sfno_instance_stats),
    ("gc_gather_gemm(Tensor[] src, Tensor?[] idx, int[] width, Tensor w, int w_plane, int ldw, Tensor bias, Tensor(a!) out, int M, int N, int act, "
     "Tensor? kscale, Tensor? kshift) -> ()", _gc_gather_gemm),
    ("gc_linear_layer_norm(Tensor a, int lda, int K, Tensor w, int w_plane, int ldw, Tensor bias, Tensor gamma, Tensor beta, Tensor? res, Tensor(a!) out, int rows) -> ()",
     _gc_linear_layer_norm),
    ("gc_sum_linear_layer_norm(Tensor[] src, int[] src_off, int[] ld, Tensor?[] idx, int K, int act, Tensor w, int w_plane, int ldw, Tensor? bias, Tensor gamma, "
     "Tensor beta, Tensor? res, Tensor(a!) out, int rows, int group=0) -> ()", _gc_sum_linear_layer_norm),
    ("gc_layer_norm(Tensor x, Tensor gamma, Tensor beta, Tensor? res, Tensor(a!) out, int rows, int N) -> ()", _gc_layer_norm),
    ("gc_segment_sum(Tensor e, Tensor offsets, Tensor(a!) out, Tensor(b!)? acc, int n_nodes, int N) -> ()", _gc_segment_sum),
    ("gc_edge_update(Tensor e_in, Tensor(c!)? e_out, Tensor[] term, int[] term_off, int[] ld, Tensor[] idx, Tensor recv, Tensor? w1f, Tensor w2f, Tensor b2, "
     "Tensor gamma, Tensor beta, Tensor(a!) agg, Tensor(b!)? heads, int rows, Tensor(d!)? probe=None, int w1_planes=2) -> ()", _gc_edge_update),
    ("gc_segment_fixup(Tensor(a!) agg, Tensor heads, Tensor nodes, Tensor first, Tensor tiles) -> ()", _gc_segment_fixup),
    ("gc_node_mlp(Tensor[] src, int[] src_off, int[] ld, Tensor w1f, Tensor w2f, Tensor b1, Tensor b2, Tensor gamma, Tensor beta, Tensor? res, int res_off, "
     "int ld_res, Tensor(a!) out, int out_off, int ld_out, int rows) -> ()", _gc_node_mlp),
]
OP_NAMES = [s.split("(", 1)[0] for s, _ in _SCHEMAS]


def register() -> None:
    """Define the schemas and attach the CUDA (ROCm) implementations.  Idempotent; importing this module registers."""
    global _registered
    if _registered:
        return
    for schema, fn in _SCHEMAS:
        _lib.define(schema)
        _lib.impl(schema.split("(", 1)[0], fn, "CUDA")
    _registered = True


register()
hip = getattr(torch.ops, _NS)
