"""GraphCast 6-h step on one MI355X: the host owns the graph, the buffers and the call order; every FLOP runs in the HIP
kernels of include/skyrim_graphcast.h and include/skyrim_sfno.h (ctypes; PyTorch is device memory + streams).

Latents are fp32 row-major ``[rows][latent]`` (grid nodes, mesh nodes, and the three edge sets).  An MLP is

    skgc_gather_gemm (gather + concat + Linear + swish)  ->  sksfno_gemm_run (Linear)  ->  skgc_layer_norm (+ residual)

and the receiver sum between an edge update and its node update is skgc_segment_sum over edges sorted by receiver.  The
embeddings of the structural features (mesh nodes, all three edge sets) do not depend on the input and are computed once in
``load_params``.  The grid-node features are read in place from a ``[186][n_grid]`` stack (two states, forcings, static
fields, structural features) with the state normalisation as a per-k affine in the loader; the output layer writes
``x(t) + diff_std * residual`` straight into the ``(83, n_lat, n_lon)`` result.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes
import os
from pathlib import Path

import numpy as np
import torch

from .. import ops
from ..sfno import engine as _sf
from . import fused as _fz
from .mesh import renumber_mesh, spatial_order, GraphStructure, build_graph, grouped_rows_by3, latitude_band, shard_graph
from .spec import N_FORCING, N_STATIC, GraphcastConfig, mlp_names, param_spec

_LIB_PATH = Path(__file__).resolve().parent.parent / "lib" / "libskyrim_graphcast.so"
EXPORTS = ["skgc_abi_version", "skgc_gather_gemm", "skgc_layer_norm", "skgc_segment_sum", "skgc_prepare_weight_perm8",
           "skgc_linear_layer_norm", "skgc_sum_linear_layer_norm", "skgc_edge_update", "skgc_segment_fixup", "skgc_node_mlp"]


class GatherDesc(ctypes.Structure):
    _fields_ = [("src", ctypes.c_void_p * 3), ("idx", ctypes.c_void_p * 3), ("ld", ctypes.c_longlong * 3), ("width", ctypes.c_int * 3),
                ("n_src", ctypes.c_int), ("kscale", ctypes.c_void_p), ("kshift", ctypes.c_void_p),
                ("w", ctypes.c_void_p), ("w_plane", ctypes.c_longlong), ("ldw", ctypes.c_int), ("bias", ctypes.c_void_p),
                ("out", ctypes.c_void_p), ("ldo", ctypes.c_longlong), ("M", ctypes.c_int), ("N", ctypes.c_int), ("act", ctypes.c_int)]


class SumDesc(ctypes.Structure):
    _fields_ = [("src", ctypes.c_void_p * 3), ("idx", ctypes.c_void_p * 3), ("ld", ctypes.c_longlong * 3), ("n_src", ctypes.c_int), ("K", ctypes.c_int),
                ("act", ctypes.c_int), ("w", ctypes.c_void_p), ("w_plane", ctypes.c_longlong), ("ldw", ctypes.c_int), ("bias", ctypes.c_void_p),
                ("gamma", ctypes.c_void_p), ("beta", ctypes.c_void_p), ("res", ctypes.c_void_p), ("out", ctypes.c_void_p), ("rows", ctypes.c_longlong),
                ("group", ctypes.c_int)]


class EdgeDesc(ctypes.Structure):
    _fields_ = [("e_in", ctypes.c_void_p), ("e_out", ctypes.c_void_p), ("term", ctypes.c_void_p * 2), ("idx", ctypes.c_void_p * 2), ("ld", ctypes.c_longlong * 2),
                ("n_term", ctypes.c_int), ("recv", ctypes.c_void_p), ("w1f", ctypes.c_void_p), ("w2f", ctypes.c_void_p), ("b2", ctypes.c_void_p),
                ("gamma", ctypes.c_void_p), ("beta", ctypes.c_void_p), ("agg", ctypes.c_void_p), ("heads", ctypes.c_void_p), ("rows", ctypes.c_longlong),
                ("has_fc1", ctypes.c_int), ("w1_planes", ctypes.c_int), ("probe", ctypes.c_void_p)]


class NodeDesc(ctypes.Structure):
    _fields_ = [("src", ctypes.c_void_p * 2), ("ld", ctypes.c_longlong * 2), ("n_src", ctypes.c_int), ("w1f", ctypes.c_void_p), ("w2f", ctypes.c_void_p),
                ("b1", ctypes.c_void_p), ("b2", ctypes.c_void_p), ("gamma", ctypes.c_void_p), ("beta", ctypes.c_void_p), ("res", ctypes.c_void_p),
                ("ld_res", ctypes.c_longlong), ("out", ctypes.c_void_p), ("ld_out", ctypes.c_longlong), ("rows", ctypes.c_longlong)]


_lib = None
ABI_VERSION = 5            # include/skyrim_graphcast.h SKGC_ABI_VERSION: 5 = K-outer w1f of skgc_node_mlp


def load_library():
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("SKYRIM_GRAPHCAST_LIB", str(_LIB_PATH))
    if not os.path.exists(path):
        raise RuntimeError(f"{path} not found: build the HIP library first (python -c 'import __graft_entry__ as g; g.build()')")
    lib = ctypes.CDLL(path)
    lib.skgc_gather_gemm.argtypes = [ctypes.POINTER(GatherDesc), ctypes.c_void_p]
    lib.skgc_layer_norm.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p]
    lib.skgc_segment_sum.argtypes = [ctypes.c_void_p] * 4 + [ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    lib.skgc_prepare_weight_perm8.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p]
    lib.skgc_linear_layer_norm.argtypes = [ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int] + \
        [ctypes.c_void_p] * 5 + [ctypes.c_longlong, ctypes.c_void_p]
    lib.skgc_sum_linear_layer_norm.argtypes = [ctypes.POINTER(SumDesc), ctypes.c_void_p]
    lib.skgc_edge_update.argtypes = [ctypes.POINTER(EdgeDesc), ctypes.c_void_p]
    lib.skgc_segment_fixup.argtypes = [ctypes.c_void_p] * 5 + [ctypes.c_int, ctypes.c_void_p]
    lib.skgc_node_mlp.argtypes = [ctypes.POINTER(NodeDesc), ctypes.c_void_p]
    for name in EXPORTS:
        getattr(lib, name).restype = ctypes.c_int
    # the fragment orders this module packs (fused.py) belong to ONE generation of kernels: a stale build, or a SKYRIM_GRAPHCAST_LIB variant
    # from other sources, would read them as something else and produce silently wrong node updates
    if lib.skgc_abi_version() != ABI_VERSION:
        raise RuntimeError(f"{path}: skgc ABI {lib.skgc_abi_version()}, this package packs for ABI {ABI_VERSION} (include/skyrim_graphcast.h); rebuild the library")
    _lib = lib
    return lib


def _check(code: int, what: str):
    if code != 0:
        raise RuntimeError(f"{what} failed with code {code}")


class GraphcastEngine:
    def __init__(self, cfg: GraphcastConfig | None = None, device: str | torch.device = "cuda:0", graph: GraphStructure | None = None,
                 shard: tuple[int, int] = (0, 1), reduce_fn=None, gather_fn=None):
        """``shard = (rank, world)``: one forecast over ``world`` GPUs (BASELINE configs[3]).  GRID side: this engine owns a latitude
        band of the grid (states, forcings and outputs have ``lat1 - lat0`` rows), with the grid->mesh edges its points send and the
        mesh->grid edges they receive.  MESH side ("all-gather of the small mesh latent"): it owns a contiguous range of mesh
        nodes with the multi-mesh edges they RECEIVE (owner computes: edge update, receiver sum, node update), and after each of the 16
        processor layers the ranks all-gather the updated node latents (n_mesh x latent fp32 = 84 MB at full size; 0.27 ms per layer
        over one xGMI link at 2 GPUs) -- ``gather_fn(out [world, per, latent], mine [per, latent])`` with ``mine`` a view of ``out[rank]`` (in place), default
        ``torch.distributed.all_gather_into_tensor``.  The other exchange of a step
        is ``reduce_fn(agg)`` = the sum over ranks of the (n_mesh, latent) aggregate of the grid->mesh messages (default:
        ``torch.distributed.all_reduce`` = RCCL over xGMI; 84 MB at full size).  ``graph``: the FULL graph (built if omitted)."""
        self.cfg = cfg or GraphcastConfig()
        self.rank, self.world = shard
        if not (0 <= self.rank < self.world <= self.cfg.n_lat):
            raise ValueError("shard = (rank, world) with 0 <= rank < world <= n_lat")
        self.reduce_fn = reduce_fn
        self.gather_fn = gather_fn
        # SKGC_EXERCISE_COLLECTIVES=1: run the two exchanges of the sharded step even with ONE rank (the collectives then copy a rank's data onto
        # itself): the cheapest proof that the default torch.distributed / RCCL branch loads and runs on a 1-GPU box (tests/test_rccl_gpu.py)
        self.exercise = bool(os.environ.get("SKGC_EXERCISE_COLLECTIVES")) and self.world == 1
        self.shard_mesh = (self.world > 1 or self.exercise) and not os.environ.get("SKGC_REPLICATED_MESH")
        if not torch.cuda.is_available():
            raise RuntimeError("GraphcastEngine needs an MI355X: the GraphCast path has no CPU fallback")
        if self.cfg.latent % 8 != 0:
            raise ValueError("latent must be a multiple of 8")
        self.lib = load_library()
        self.sf = _sf.load_library()
        self.device = torch.device(device)
        full = graph or build_graph(self.cfg.n_lat, self.cfg.n_lon, self.cfg.splits)
        # The multi-mesh numbers its nodes level by level; the engine works on a spatially coherent numbering (Morton order of the node
        # positions) so that the sender rows gathered for neighbouring receivers -- neighbouring tiles of the receiver-sorted edge list -- are
        # neighbours in memory too.  Mesh nodes never leave the engine: invisible outside (SKGC_MESH_ORDER=level keeps the caller's numbering).
        if os.environ.get("SKGC_MESH_ORDER", "spatial") != "level":
            full = renumber_mesh(full, spatial_order(full.mesh_pos))
        self.lat0, self.lat1 = latitude_band(self.cfg.n_lat, self.rank, self.world)
        self.graph = full if self.world == 1 else shard_graph(full, self.cfg.n_lat, self.cfg.n_lon, self.rank, self.world)
        self.prepared = False
        self.profiling = False
        self._events = []
        self.fused_ln = self.cfg.latent == 512 and not os.environ.get("SKGC_UNFUSED_LN")
        # edge MLPs by distributivity (fc1 of concat(e, v_s[send], v_r[recv]) = e W_e^T + (v_s W_s^T)[send] + (v_r W_r^T)[recv]): the node
        # terms once per node, the edge term once per layer (once per model for the encoder / decoder, whose edge latents are
        # input-independent); needs the fused Linear + LayerNorm kernel (latent 512)
        self.split_edges = self.fused_ln and not os.environ.get("SKGC_CONCAT_EDGES")
        # round 4: every interaction-network update as ONE kernel (csrc/graphcast_fused.hip): edge update + receiver sum on packed rows with
        # one-plane fp16 edge operands, node updates on fp32 rows with three MFMA terms.  SKGC_UNFUSED=1 keeps the round-3 kernel sequence.
        self.fused = self.split_edges and not os.environ.get("SKGC_UNFUSED")
        # planes of the processor edge MLPs' first Linear (edge part): 1 = W_e as one fp16 plane (one MFMA term: +1.4e-4 of the predicted
        # increment at production width and depth, tools/graphcast_numerics.py), 2 = hi/lo planes (two terms)
        self.w1_planes = int(os.environ.get("SKGC_W1_PLANES", "1"))
        if self.w1_planes not in (1, 2):
            raise ValueError("SKGC_W1_PLANES is 1 or 2")
        self.state_shape = (self.cfg.n_vars, self.lat1 - self.lat0, self.cfg.n_lon)

    def release(self):
        """Drop the packed graph, every prepared weight and all latent / work buffers (~47 GB at 721x1440; GlobalModel.release_model).  The C ABI
        holds no state of its own: all device memory is torch tensors owned here.  The engine is unusable afterwards."""
        keep = ("cfg", "rank", "world", "reduce_fn", "gather_fn", "lib", "sf", "device", "state_shape")
        kept = {k: v for k, v in vars(self).items() if k in keep}
        self.__dict__.clear()
        self.__dict__.update(kept)
        self.prepared = False
        self._events = []

    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ---- profiling (same contract as SfnoEngine) ---------------------------------------------------- #
    def _mark(self, label: str, flops: float = 0.0, terms: float = 3.0):
        """flops: dense FLOPs of the launch that follows; terms: MFMA terms it executes per product (3 = fp16 hi/lo on both operands)."""
        if self.profiling:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream(self.device))
            self._events.append((label, ev, flops, flops * terms))

    def profile_read(self) -> list[dict]:
        torch.cuda.synchronize(self.device)
        out: dict[str, dict] = {}
        for (label, ev, fl, mf), (_, nxt, _, _) in zip(self._events[:-1], self._events[1:]):
            if label == "end":
                continue
            d = out.setdefault(label, dict(name=label, launches=0, total_ms=0.0, flops=0.0, mfma_flops=0.0))
            d["launches"] += 1
            d["total_ms"] += ev.elapsed_time(nxt)
            d["flops"] += fl
            d["mfma_flops"] += mf
        self._events = []
        return list(out.values())

    # ---- kernels ------------------------------------------------------------------------------------- #
    def _fc1(self, W, bias, sources, rows, out, kscale=None, kshift=None, label="mlp"):
        """sources: [(tensor [n][ld], index tensor or None, width)] -> out[rows][latent] = swish(concat(...) W^T + b)."""
        self._mark(label, 2.0 * rows * W.N * W.K)
        ops.hip.gc_gather_gemm([t for t, _, _ in sources], [i for _, i, _ in sources], [w for _, _, w in sources], W.buf, W.plane, W.ldw, bias, out,
                               rows, W.N, 2, kscale, kshift)

    def _gemm(self, a, W, out, M, *, a_sm, a_sk, o_sm, o_sn, bias=None, res_post=None, act=0, kscale=None, kshift=None, label="mlp"):
        self._mark(label, 2.0 * M * W.N * W.K)
        big = 1 << 30
        geom = [0, 0, big, a_sm, 0, a_sk, 0, W.plane, W.ldw, 0, 0, big, o_sm, 0, o_sn, M, W.N, W.K, 1, act, 0, 0, 0, 0, 0, 3]
        ops.hip.sfno_gemm(a, W.buf, out, bias, None, res_post, kscale, kshift, None, geom)

    def _ln(self, x, g, b, res, out, rows, label="layer_norm"):
        self._mark(label)
        ops.hip.gc_layer_norm(x, g, b, res, out, rows, self.cfg.latent)

    def _segsum(self, e, offsets, out, n_nodes, acc=None):
        self._mark("segment_sum")
        ops.hip.gc_segment_sum(e, offsets, out, acc, n_nodes, self.cfg.latent)

    def _mlp(self, name, sources, rows, out, res=None, label=None):
        """out = (res +) LayerNorm(fc2(swish(fc1(concat(sources)))));  out may alias res."""
        m = self.m[name]
        L = self.cfg.latent
        self._fc1(m["fc1"], m["b1"], sources, rows, self.b_h, label=label or name.split(".")[0])
        if m.get("fc2p") is not None:                # latent 512: second Linear + LayerNorm (+ residual) in one kernel
            buf, plane, ldw = m["fc2p"]
            self._mark(label or name.split(".")[0], 2.0 * rows * L * L)
            ops.hip.gc_linear_layer_norm(self.b_h, L, L, buf, plane, ldw, m["b2"], m["g"], m["b"], res, out, rows)
            return
        self._gemm(self.b_h, m["fc2"], self.b_t, rows, a_sm=L, a_sk=1, o_sm=L, o_sn=1, bias=m["b2"], label=label or name.split(".")[0])
        self._ln(self.b_t, m["g"], m["b"], res, out, rows)

    def _sum_ln(self, name, sources, rows, out, res=None, label=None, group=0):
        """out = (res +) LayerNorm(fc2(swish(sum of the gathered source rows))) -- sources: [(flat tensor, element offset, ld, index or None)].
        group = 3: rows counts groups of three (virtual row order, see load_params) and out[g] is the sum over the group."""
        m, L = self.m[name], self.cfg.latent
        buf, plane, ldw = m["fc2p"]
        self._mark(label or name.split(".")[0], 2.0 * rows * (group or 1) * L * L)
        ops.hip.gc_sum_linear_layer_norm([t for t, _, _, _ in sources], [o for _, o, _, _ in sources], [d for _, _, d, _ in sources], [i for _, _, _, i in sources],
                                         L, 2, buf, plane, ldw, m["b2"], m["g"], m["b"], res, out, rows, group)

    def _edge_mlp(self, name, edge_term, vs, idx_s, vr, idx_r, rows, out, res=None, label=None):
        """Edge update by distributivity.  edge_term: a precomputed [rows][L] tensor (e W_e^T + b1, input-independent edge latents) or the
        current edge latents (then e W_e^T + b1 is one GEMM); vs / vr: node latents whose W_s / W_r images are gathered (vr None: already
        folded into edge_term)."""
        m, L = self.m[name], self.cfg.latent
        lab = label or name.split(".")[0]
        if m.get("static_edge"):
            E = edge_term
        else:
            self._gemm(edge_term, m["w_e"], self.b_h, rows, a_sm=L, a_sk=1, o_sm=L, o_sn=1, bias=m["b1"], label=lab)
            E = self.b_h
        srcs = [(E, 0, L, None)]
        if vr is not None and vs is vr:                  # processor: one GEMM for both node terms, [n][2L] = v [W_s; W_r]^T
            n = vs.shape[0]
            self._gemm(vs, m["w_sr"], self.b_ps, n, a_sm=L, a_sk=1, o_sm=2 * L, o_sn=1, label=lab)
            srcs += [(self.b_ps, 0, 2 * L, idx_s), (self.b_ps, L, 2 * L, idx_r)]
        else:
            ns = vs.shape[0]
            self._gemm(vs, m["w_s"], self.b_ps, ns, a_sm=L, a_sk=1, o_sm=L, o_sn=1, label=lab)
            srcs.append((self.b_ps, 0, L, idx_s))
            if vr is not None:
                self._gemm(vr, m["w_r"], self.b_pr, vr.shape[0], a_sm=L, a_sk=1, o_sm=L, o_sn=1, label=lab)
                srcs.append((self.b_pr, 0, L, idx_r))
        self._sum_ln(name, srcs, rows, out, res=res, label=lab)

    # ---- fused updates (round 4) ------------------------------------------------------------------------- #
    def _edge_update(self, f, e_in, e_out, terms, agg, label):
        """f: a prepared edge set (``_prepare_fused``); terms: [(tensor, element offset, ld, index tensor)]."""
        L = self.cfg.latent
        fc1 = f.get("w1f") is not None
        self._mark(label, 2.0 * f["n_edges"] * L * L * (2 if fc1 else 1), (2.0 + self.w1_planes) / 2.0 if fc1 else 2.0)
        ops.hip.gc_edge_update(e_in, e_out, [t for t, _, _, _ in terms], [o for _, o, _, _ in terms], [d for _, _, d, _ in terms], [i for _, _, _, i in terms],
                               f["recv"], f.get("w1f"), f["w2f"], f["b2"], f["g"], f["b"], agg, f.get("heads"), f["rows"], None, self.w1_planes)
        if f.get("fix") is not None:
            self._mark(label)
            ops.hip.gc_segment_fixup(agg, f["heads"], *f["fix"])

    def _node_mlp(self, name, srcs, rows, res, out, label):
        """srcs / res / out: (tensor, element offset, ld).  out = res + LayerNorm(fc2(swish(fc1(concat(srcs)))))."""
        m, L = self.m[name], self.cfg.latent
        self._mark(label, 2.0 * rows * L * L * (len(srcs) + 1))
        ops.hip.gc_node_mlp([t for t, _, _ in srcs], [o for _, o, _ in srcs], [d for _, _, d in srcs], m["w1f"], m["w2f"], m["b1"], m["b2"], m["g"], m["b"],
                            res[0] if res is not None else None, res[1] if res is not None else 0, res[2] if res is not None else L, out[0], out[1], out[2], rows)

    def _prepare_fused(self, p):
        """Packed row orders, blocked fp16 edge operands, fragment-order weights and "pos"-ordered node-term weights (fused.py)."""
        c, g, dev, L = self.cfg, self.graph, self.device, self.cfg.latent
        f32 = lambda t: t.float().contiguous().to(dev)  # noqa: E731
        i32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(dev)  # noqa: E731
        weng = type("W", (), {"device": dev, "lib": self.sf, "_stream": self._stream})()
        upos = torch.from_numpy(_fz.unit_at_pos(L))

        def pack(edges):
            row_edge = _fz.pack_segments(edges[:, 1])
            ok = row_edge >= 0
            re = np.where(ok, row_edge, 0)
            recv = np.where(ok, edges[re, 1], -1)
            send = np.where(ok, edges[re, 0], -1)
            d = dict(rows=len(row_edge), n_edges=len(edges), row_edge=row_edge, recv=i32(recv), send=i32(send))
            nodes, first, tiles = _fz.continuation_list(recv)
            # the heads buffer ALWAYS (2 KB per 128-row tile): continuation_list has just proven on the host whether a run crosses a tile
            # boundary, and an op called without the buffer has to find that out on the device -- one blocking .item() per edge update,
            # 17 per step on graphs without continuations (small meshes), which also made the step uncapturable as a HIP graph
            d["heads"] = torch.zeros(len(row_edge) // _fz.TILE, L, dtype=torch.float32, device=dev)
            if len(nodes):
                d["fix"] = (i32(nodes), i32(first), i32(tiles))
            return d

        def blocked(x, row_edge, cols=None):
            """x [E][L] fp32 on the device -> the packed rows as one blocked fp16 plane (padding rows zero, columns optionally permuted)."""
            out = torch.empty(len(row_edge) * L, dtype=torch.float16, device=dev)
            re = torch.from_numpy(row_edge.astype(np.int64)).to(dev)
            for r0 in range(0, len(row_edge), 1 << 19):
                ix = re[r0:r0 + (1 << 19)]
                v = x[ix.clamp_min(0)]
                v[ix < 0] = 0.0
                if cols is not None:
                    v = v[:, cols]
                out[r0 * L:(r0 + len(ix)) * L] = _fz.to_blocked_f16(v)
            return out

        def mlp_frags(name, k0=None, k1=None):
            w1 = p[name + ".fc1.weight"]
            w2 = p[name + ".fc2.weight"]
            d = dict(w2f=_fz.prep_w2_fragments(f32(w2)), b2=self.m[name]["b2"], g=self.m[name]["g"], b=self.m[name]["b"])
            if k0 is not None:
                d["w1f"] = _fz.prep_w1_fragments(f32(w1[:, k0:k1]), self.w1_planes)
            return d

        cols = upos.to(dev)
        F = {}
        # grid -> mesh: the prepared term (e1_0 already holds e W_e^T + b1 + (vm0 W_r^T)[recv]) in "pos" columns; sender term per grid node
        F["g2m"] = pack(g.g2m_edges)
        F["g2m"].update(mlp_frags("g2m.edge"))
        F["g2m"]["static"] = blocked(self.e1_0, F["g2m"]["row_edge"], cols)
        w1 = p["g2m.edge.fc1.weight"]
        F["g2m"]["w_s"] = _sf._Weight(weng, w1[:, L:2 * L][upos])
        # multi-mesh: edge latents live in the packed order as one fp16 plane; [W_s; W_r] in "pos" rows with b1 folded into the sender term
        me = g.mesh_edges[self.me0:self.me1]
        F["mesh"] = pack(me)
        F["mesh"]["em0"] = blocked(self.em_0, F["mesh"]["row_edge"])
        F["mesh"]["em"] = torch.zeros_like(F["mesh"]["em0"])
        deg = np.bincount(me[:, 1] - self.mn0, minlength=self.mn1 - self.mn0) if len(me) else np.zeros(self.mn1 - self.mn0, dtype=np.int64)
        F["mesh"]["zero_agg"] = bool((deg == 0).any())
        for i in range(c.steps):
            name = f"proc.{i}.edge"
            w1, b1 = p[name + ".fc1.weight"], p[name + ".fc1.bias"]
            d = mlp_frags(name, 0, L)
            d["w_sr"] = _sf._Weight(weng, torch.cat([w1[:, L:2 * L][upos], w1[:, 2 * L:][upos]], dim=0))
            d["b_sr"] = f32(torch.cat([b1[upos], torch.zeros(L, dtype=b1.dtype)]))
            F[name] = d
        # mesh -> grid
        F["m2g"] = pack(g.m2g_edges)
        F["m2g"].update(mlp_frags("m2g.edge"))
        F["m2g"]["static"] = blocked(self.e2_0, F["m2g"]["row_edge"], cols)
        w1 = p["m2g.edge.fc1.weight"]
        F["m2g"]["w_s"], F["m2g"]["w_r"] = _sf._Weight(weng, w1[:, L:2 * L][upos]), _sf._Weight(weng, w1[:, 2 * L:][upos])
        F["m2g"]["zero_agg"] = bool((np.bincount(g.m2g_edges[:, 1], minlength=g.n_grid) == 0).any())
        # node updates: fragment-order weights, three-term kernel
        for name in ["g2m.mesh_node", "g2m.grid_node", "m2g.grid_node"] + [f"proc.{i}.node" for i in range(c.steps)]:
            self.m[name]["w1f"] = _fz.prep_w1_node(f32(p[name + ".fc1.weight"]))
            self.m[name]["w2f"] = _fz.prep_w2_fragments(f32(p[name + ".fc2.weight"]))
        self.F = F
        del self.e1_0, self.e2_0
        torch.cuda.current_stream(dev).synchronize()

    def _step_fused(self, y):
        """The graph network of one step on the fused kernels: 3 launches per processor layer (node terms, edge update + receiver sum,
        node update)."""
        c, L, g, F = self.cfg, self.cfg.latent, self.graph, self.F
        P, V = self.P, c.n_vars
        # encoder: grid -> mesh
        f = F["g2m"]
        self._gemm(self.vg, f["w_s"], self.b_ps, P, a_sm=L, a_sk=1, o_sm=L, o_sn=1, label="encoder")
        self.agg_m.zero_()                                   # receivers without an edge on this rank (grid-sharded runs) must read zero
        self._edge_update(f, f["static"], None, [(self.b_ps, 0, L, f["send"])], self.agg_m, "encoder")
        if self.world > 1 or self.exercise:
            self._mark("exchange")
            if self.reduce_fn is not None:
                self.reduce_fn(self.agg_m)
            else:
                import torch.distributed as dist
                dist.all_reduce(self.agg_m, op=dist.ReduceOp.SUM)
        self._node_mlp("g2m.mesh_node", [(self.vm0, 0, L), (self.agg_m, 0, L)], g.n_mesh, (self.vm0, 0, L), (self.vm, 0, L), "encoder")
        self._node_mlp("g2m.grid_node", [(self.vg, 0, L)], P, (self.vg, 0, L), (self.vg, 0, L), "encoder")
        # processor
        nl, o0 = self.mn1 - self.mn0, self.mn0 * L
        f = F["mesh"]
        for i in range(c.steps):
            d = F[f"proc.{i}.edge"]
            self._gemm(self.vm, d["w_sr"], self.b_ps, g.n_mesh, a_sm=L, a_sk=1, o_sm=2 * L, o_sn=1, bias=d["b_sr"], label="processor")
            if f["zero_agg"]:
                self.agg_m[self.mn0:self.mn1].zero_()
            self._edge_update({**f, **d}, f["em0"] if i == 0 else f["em"], f["em"], [(self.b_ps, 0, 2 * L, f["send"]), (self.b_ps, L, 2 * L, f["recv"])], self.agg_m, "processor")
            self._node_mlp(f"proc.{i}.node", [(self.vm, o0, L), (self.agg_m, o0, L)], nl, (self.vm, o0, L), (self.vm, o0, L), "processor")
            if self.shard_mesh:
                self._exchange_nodes(nl)
        # decoder: mesh -> grid
        f = F["m2g"]
        self._gemm(self.vm, f["w_s"], self.b_ps, g.n_mesh, a_sm=L, a_sk=1, o_sm=L, o_sn=1, label="decoder")
        self._gemm(self.vg, f["w_r"], self.b_pr, P, a_sm=L, a_sk=1, o_sm=L, o_sn=1, label="decoder")
        if f["zero_agg"]:
            self.agg_g.zero_()
        self._edge_update(f, f["static"], None, [(self.b_ps, 0, L, f["send"]), (self.b_pr, 0, L, f["recv"])], self.agg_g, "decoder")
        self._node_mlp("m2g.grid_node", [(self.vg, 0, L), (self.agg_g, 0, L)], P, (self.vg, 0, L), (self.vg, 0, L), "decoder")

    def _exchange_nodes(self, nl):
        """all-gather of the updated node latents of a mesh-sharded step (84 MB at full size)."""
        self._mark("exchange")
        # IN PLACE: the node latents live in a buffer of world x mn_per rows (the last rank's share padded), every rank's share at its own
        # offset, so the collective's output IS ``vm`` and its input is this rank's slice of it (ncclAllGather's in-place form:
        # sendbuff = recvbuff + rank x count) -- no staging copy, no scatter loop
        mine = self.vm_store[self.rank * self.mn_per:(self.rank + 1) * self.mn_per]
        if self.gather_fn is not None:
            self.gather_fn(self.vm_store.view(self.world, self.mn_per, self.cfg.latent), mine)
        else:
            import torch.distributed as dist
            dist.all_gather_into_tensor(self.vm_store, mine)

    # ---- prepare ------------------------------------------------------------------------------------- #
    def load_params(self, params: dict):
        c, g, dev = self.cfg, self.graph, self.device
        for name, shape in param_spec(c):
            if name not in params or tuple(params[name].shape) != tuple(shape):
                raise ValueError(f"parameter {name}: expected shape {shape}, got {tuple(params[name].shape) if name in params else None}")
        L = c.latent
        with torch.cuda.device(dev):
            f32 = lambda t: t.float().contiguous().to(dev)  # noqa: E731
            p = {k: v.double() for k, v in params.items() if k != "static"}
            mean, std, dstd = p["norm.mean"], p["norm.std"], p["norm.diff_std"]
            weng = type("W", (), {"device": dev, "lib": self.sf, "_stream": self._stream})()       # what sfno's _Weight needs
            self.m = {}
            for name, d_in, d_out, ln in mlp_names(c):
                w2, b2 = p[name + ".fc2.weight"], p[name + ".fc2.bias"]
                if name == "out":                                       # fold the residual's de-normalisation into the last layer
                    w2, b2 = w2 * dstd[:, None], b2 * dstd
                self.m[name] = dict(fc1=_sf._Weight(weng, p[name + ".fc1.weight"]), b1=f32(p[name + ".fc1.bias"]),
                                    fc2=_sf._Weight(weng, w2), b2=f32(b2),
                                    g=f32(p[name + ".ln.weight"]) if ln else None, b=f32(p[name + ".ln.bias"]) if ln else None)
                if self.split_edges and name.endswith(".edge") and not name.startswith("embed"):
                    w1 = p[name + ".fc1.weight"]
                    d = self.m[name]
                    d["w_e"] = _sf._Weight(weng, w1[:, :L])
                    if name.startswith("proc."):
                        d["w_sr"] = _sf._Weight(weng, torch.cat([w1[:, L:2 * L], w1[:, 2 * L:]], dim=0))
                    else:
                        d["w_s"], d["w_r"] = _sf._Weight(weng, w1[:, L:2 * L]), _sf._Weight(weng, w1[:, 2 * L:])
                if ln and self.fused_ln:                                 # perm8 copy of the second Linear for the fused Linear + LayerNorm kernel
                    src = f32(w2)
                    planes = torch.empty(2 * L * L, dtype=torch.float16, device=dev)
                    _check(self.lib.skgc_prepare_weight_perm8(src.data_ptr(), L, L, planes.data_ptr(), L * L, L, self._stream()), "skgc_prepare_weight_perm8")
                    torch.cuda.current_stream(dev).synchronize()
                    self.m[name]["fc2p"] = (planes, L * L, L)
            n_state = 2 * c.n_vars
            one, zero = torch.ones(N_FORCING + N_STATIC + 3, dtype=torch.float64), torch.zeros(N_FORCING + N_STATIC + 3, dtype=torch.float64)
            self.in_scale = f32(torch.cat([1.0 / std, 1.0 / std, one]))
            self.in_shift = f32(torch.cat([-mean / std, -mean / std, zero]))
            i32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(dev)  # noqa: E731

            def csr(edges, n):
                off = np.zeros(n + 1, dtype=np.int64)
                np.add.at(off, edges[:, 1] + 1, 1)
                return i32(np.cumsum(off))

            self.g2m_s, self.g2m_r, self.g2m_off = i32(g.g2m_edges[:, 0]), i32(g.g2m_edges[:, 1]), csr(g.g2m_edges, g.n_mesh)
            self.me_s, self.me_r, self.me_off = i32(g.mesh_edges[:, 0]), i32(g.mesh_edges[:, 1]), csr(g.mesh_edges, g.n_mesh)
            self.m2g_s, self.m2g_r, self.m2g_off = i32(g.m2g_edges[:, 0]), i32(g.m2g_edges[:, 1]), csr(g.m2g_edges, g.n_grid)
            P, E1, EM, E2 = g.n_grid, len(g.g2m_edges), len(g.mesh_edges), len(g.m2g_edges)
            # mesh-side ownership: nodes [mn0, mn1) and the multi-mesh edges they receive (edges are sorted by receiver: one slice)
            self.mn_per = (g.n_mesh + self.world - 1) // self.world
            self.mn0, self.mn1 = (min(self.rank * self.mn_per, g.n_mesh), min((self.rank + 1) * self.mn_per, g.n_mesh)) if self.shard_mesh else (0, g.n_mesh)
            off = np.zeros(g.n_mesh + 1, dtype=np.int64)
            np.add.at(off, g.mesh_edges[:, 1] + 1, 1)
            off = np.cumsum(off)
            self.me0, self.me1 = int(off[self.mn0]), int(off[self.mn1])
            if self.shard_mesh:
                self.me_s, self.me_r = self.me_s[self.me0:self.me1].contiguous(), self.me_r[self.me0:self.me1].contiguous()
                self.me_off = i32(off[self.mn0:self.mn1 + 1] - self.me0)
                EM = self.me1 - self.me0
            self.P, self.E1, self.EM, self.E2 = P, E1, EM, E2
            # mesh -> grid: every grid node receives exactly three edges (the corners of its triangle), stored receiver by receiver.
            # The edge MLP then runs in "virtual row" order -- row 48 t + 16 a + l = edge a of grid node 16 t + l -- so that its epilogue
            # sums the three LayerNorm outputs of a node in registers: the updated edge latents and the receiver sum never reach HBM
            self.m2g_group = None
            r2 = g.m2g_edges[:, 1]
            if self.split_edges and os.environ.get("SKGC_M2G_SEGSUM", "0") != "1" and E2 == 3 * P and np.array_equal(r2, np.repeat(np.arange(P), 3)):
                edge = grouped_rows_by3(P)
                self.m2g_group = (i32(edge), i32(g.m2g_edges[edge, 0]), i32(r2[edge]))
            buf = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)  # noqa: E731
            rows_max = max(P, E1, E2, EM, g.n_mesh)
            self.b_h, self.b_t = buf(rows_max, L), buf(rows_max, L)
            self.feat = torch.zeros(c.grid_in, P, dtype=torch.float32, device=dev)          # [186][n_grid]: states, forcings, static, structural
            self.feat[n_state + N_FORCING:n_state + N_FORCING + N_STATIC] = f32(params["static"][:, self.lat0:self.lat1]).reshape(N_STATIC, P)
            self.feat[n_state + N_FORCING + N_STATIC:] = torch.from_numpy(g.grid_node_feat.T.copy()).to(dev)
            # mesh-node latents: in a mesh-sharded run the buffer holds world x mn_per rows (>= n_mesh) so that the all-gather writes it in place
            self.vm_store = torch.zeros(self.world * self.mn_per if self.shard_mesh else g.n_mesh, L, dtype=torch.float32, device=dev)
            self.vg, self.vm = buf(P, L), self.vm_store[:g.n_mesh]
            self.agg_m, self.agg_g = buf(g.n_mesh, L), buf(P, L)
            if self.fused:                                       # the fused path keeps no fp32 edge latents at all
                self.e1 = self.em = self.e2 = self.de = None
            else:
                self.e1, self.em, self.e2, self.de = buf(E1, L), buf(EM, L), buf(E2, L), buf(EM, L)
            # input-independent embeddings of the structural features
            self.vm0, self.e1_0, self.em_0, self.e2_0 = buf(g.n_mesh, L), buf(E1, L), buf(EM, L), buf(E2, L)
            for name, feat, out in (("embed.mesh", g.mesh_node_feat, self.vm0), ("embed.g2m_edge", g.g2m_edge_feat, self.e1_0),
                                    ("embed.mesh_edge", g.mesh_edge_feat[self.me0:self.me1], self.em_0), ("embed.m2g_edge", g.m2g_edge_feat, self.e2_0)):
                ft = torch.from_numpy(np.ascontiguousarray(feat)).to(dev)
                self._mlp(name, [(ft, None, ft.shape[1])], ft.shape[0], out)
            if self.split_edges:
                # input-independent parts of the encoder / decoder edge MLPs' first Linear, once per model:
                #   grid->mesh: e1_0 W_e^T + b1 + (vm0 W_r^T)[receiver]   (edge latents AND the receiving mesh latents are structural)
                #   mesh->grid: e2_0 W_e^T + b1
                self.b_ps = buf(max(P, g.n_mesh), 2 * L)
                self.b_pr = buf(max(P, g.n_mesh), L)
                m1, m2 = self.m["g2m.edge"], self.m["m2g.edge"]
                self._gemm(self.e1_0, m1["w_e"], self.b_h, E1, a_sm=L, a_sk=1, o_sm=L, o_sn=1, bias=m1["b1"])
                self._gemm(self.vm0, m1["w_r"], self.b_pr, g.n_mesh, a_sm=L, a_sk=1, o_sm=L, o_sn=1)
                torch.cuda.current_stream(dev).synchronize()
                self.e1_0.copy_(self.b_h[:E1])
                chunk = 1 << 20
                for i0 in range(0, E1, chunk):
                    self.e1_0[i0:i0 + chunk] += self.b_pr[self.g2m_r[i0:i0 + chunk].long()]
                self._gemm(self.e2_0, m2["w_e"], self.b_h, E2, a_sm=L, a_sk=1, o_sm=L, o_sn=1, bias=m2["b1"])
                torch.cuda.current_stream(dev).synchronize()
                self.e2_0.copy_(self.b_h[:E2])
                m1["static_edge"] = m2["static_edge"] = True
            if self.fused:
                self._prepare_fused(p)
            torch.cuda.current_stream(dev).synchronize()
        self.prepared = True

    # ---- step ----------------------------------------------------------------------------------------- #
    def step(self, x_prev: torch.Tensor, x_cur: torch.Tensor, forcing: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
        """(83, n_lat, n_lon) states at t-6h and t, (15, n_lat, n_lon) forcings -> state at t+6h (fp32, engine device)."""
        if not self.prepared:
            raise RuntimeError("GraphcastEngine.step before load_params: not prepared")
        c, L = self.cfg, self.cfg.latent
        fshape = (N_FORCING,) + self.state_shape[1:]
        for t, shape in ((x_prev, self.state_shape), (x_cur, self.state_shape), (forcing, fshape)):
            if t.device != self.device or t.dtype != torch.float32 or tuple(t.shape) != shape or not t.is_contiguous():
                raise ValueError(f"expected contiguous float32 tensors of shape {self.state_shape} (states) / {fshape} (forcings) on {self.device}")
        P, V = self.P, c.n_vars
        with torch.cuda.device(self.device):
            y = out if out is not None else torch.empty(self.state_shape, dtype=torch.float32, device=self.device)
            if y.device != self.device or y.dtype != torch.float32 or tuple(y.shape) != self.state_shape or not y.is_contiguous():
                raise ValueError("bad output tensor")
            self.feat[:V].copy_(x_prev.reshape(V, P))
            self.feat[V:2 * V].copy_(x_cur.reshape(V, P))
            self.feat[2 * V:2 * V + N_FORCING].copy_(forcing.reshape(N_FORCING, P))
            m = self.m["embed.grid"]
            # grid-node embedder: rows = grid nodes (contiguous), k = feature (stride n_grid), normalisation in the loader
            self._gemm(self.feat, m["fc1"], self.b_h, P, a_sm=1, a_sk=P, o_sm=L, o_sn=1, bias=m["b1"], act=2,
                       kscale=self.in_scale, kshift=self.in_shift, label="embed")
            if m.get("fc2p") is not None:            # latent 512: second Linear + LayerNorm in one kernel, the pre-norm rows stay on chip
                buf, plane, ldw = m["fc2p"]
                self._mark("embed", 2.0 * P * L * L)
                ops.hip.gc_linear_layer_norm(self.b_h, L, L, buf, plane, ldw, m["b2"], m["g"], m["b"], None, self.vg, P)
            else:
                self._gemm(self.b_h, m["fc2"], self.b_t, P, a_sm=L, a_sk=1, o_sm=L, o_sn=1, bias=m["b2"], label="embed")
                self._ln(self.b_t, m["g"], m["b"], None, self.vg, P)
            if self.fused:
                self._step_fused(y)
            else:
                # encoder: grid -> mesh
                if self.split_edges:
                    self._edge_mlp("g2m.edge", self.e1_0, self.vg, self.g2m_s, None, None, self.E1, self.e1, label="encoder")
                else:
                    self._mlp("g2m.edge", [(self.e1_0, None, L), (self.vg, self.g2m_s, L), (self.vm0, self.g2m_r, L)], self.E1, self.e1, label="encoder")
                self._segsum(self.e1, self.g2m_off, self.agg_m, self.graph.n_mesh)
                if self.world > 1 or self.exercise:      # the one exchange of a grid-sharded step: sum the partial aggregates over ranks
                    self._mark("exchange")
                    if self.reduce_fn is not None:
                        self.reduce_fn(self.agg_m)
                    else:
                        import torch.distributed as dist
                        dist.all_reduce(self.agg_m, op=dist.ReduceOp.SUM)
                self._mlp("g2m.mesh_node", [(self.vm0, None, L), (self.agg_m, None, L)], self.graph.n_mesh, self.vm, res=self.vm0, label="encoder")
                self._mlp("g2m.grid_node", [(self.vg, None, L)], P, self.vg, res=self.vg, label="encoder")
                # processor on the multi-mesh: this rank's node range and the edges it receives (everything, on one GPU)
                nl = self.mn1 - self.mn0
                vm_own, agg_own = self.vm[self.mn0:self.mn1], self.agg_m[self.mn0:self.mn1]
                self.em.copy_(self.em_0)
                for i in range(c.steps):
                    de = self.de
                    if self.split_edges:
                        self._edge_mlp(f"proc.{i}.edge", self.em, self.vm, self.me_s, self.vm, self.me_r, self.EM, de, label="processor")
                    else:
                        self._mlp(f"proc.{i}.edge", [(self.em, None, L), (self.vm, self.me_s, L), (self.vm, self.me_r, L)], self.EM, de, label="processor")
                    self._segsum(de, self.me_off, agg_own, nl, acc=self.em)                      # receiver sum; em += de rides along
                    self._mlp(f"proc.{i}.node", [(vm_own, None, L), (agg_own, None, L)], nl, vm_own, res=vm_own, label="processor")
                    if self.shard_mesh:                  # all-gather of the updated node latents (84 MB at full size)
                        self._exchange_nodes(nl)
                # decoder: mesh -> grid
                if self.m2g_group is not None:
                    # edge update + receiver sum in one kernel: node terms once per node, then sum over a node's three edges of
                    # LayerNorm(fc2(swish(e W_e + b + (v_m W_s)[sender] + (v_g W_r)[node])))
                    me, ie = self.m["m2g.edge"], self.m2g_group
                    self._gemm(self.vm, me["w_s"], self.b_ps, self.graph.n_mesh, a_sm=L, a_sk=1, o_sm=L, o_sn=1, label="decoder")
                    self._gemm(self.vg, me["w_r"], self.b_pr, P, a_sm=L, a_sk=1, o_sm=L, o_sn=1, label="decoder")
                    self._sum_ln("m2g.edge", [(self.e2_0, 0, L, ie[0]), (self.b_ps, 0, L, ie[1]), (self.b_pr, 0, L, ie[2])], P, self.agg_g, label="decoder", group=3)
                else:
                    if self.split_edges:
                        self._edge_mlp("m2g.edge", self.e2_0, self.vm, self.m2g_s, self.vg, self.m2g_r, self.E2, self.e2, label="decoder")
                    else:
                        self._mlp("m2g.edge", [(self.e2_0, None, L), (self.vm, self.m2g_s, L), (self.vg, self.m2g_r, L)], self.E2, self.e2, label="decoder")
                    self._segsum(self.e2, self.m2g_off, self.agg_g, P)
                self._mlp("m2g.grid_node", [(self.vg, None, L), (self.agg_g, None, L)], P, self.vg, res=self.vg, label="decoder")
            # output layer: x(t+6h) = x(t) + diff_std * MLP(vg), written channel-major
            mo = self.m["out"]
            self._fc1(mo["fc1"], mo["b1"], [(self.vg, None, L)], P, self.b_h, label="output")
            self._gemm(self.b_h, mo["fc2"], y, P, a_sm=L, a_sk=1, o_sm=1, o_sn=P, bias=mo["b2"], res_post=self.feat[V:2 * V], label="output")
            self._mark("end")
        return y
