"""Host side of the fused GraphCast kernels (csrc/graphcast_fused.hip): row packing, operand layouts, weight fragments.

Everything here runs once per model in ``GraphcastEngine.load_params`` (torch / numpy on whatever device the tensors live on);
the step itself only launches kernels.  The layouts are the kernel's contract and are pinned lane by lane on the CPU by
tests/test_graphcast_fused_layout.py (an emulation of the kernel's index algebra against plain matrix products).

Rows and tiles.  An edge kernel works on tiles of ``TILE`` = 128 consecutive rows of a PACKED edge order: edges stay sorted by
receiver, and a receiver's run of edges is never cut by a tile boundary unless it is longer than a tile -- a tile is padded with dummy
rows (edge id -1) instead.  The receiver sum then completes inside the workgroup that produced the rows (no atomics, no second
pass); only runs longer than 128 edges (grid->mesh edges into polar mesh nodes) continue across tiles, and their continuation
pieces go to a side buffer that ``skgc_segment_fixup`` adds in a fixed order.

Blocked fp16 matrices.  ``[R/16][K/32][16][32]``: 1 KiB blocks of 16 rows x 32 columns, so that one wave instruction (64 lanes x 16 B)
reads one contiguous block = one MFMA B-operand fragment (csrc/common.h: blk_off).

"pos" column order.  The first Linear leaves, in lane (l & 15, g = l >> 4) of a wave, the hidden units 32 j + 16 n + 4 g + r
(n < 2, r < 4) of chunk j.  Everything that is ADDED to that pre-activation (the gathered per-node terms, the prepared static edge
terms) is stored with unit u at column pos(u) = 32 j + 8 g + 4 n + r, so a lane's eight values are 32 (fp32) or 16 (fp16) contiguous bytes.
"""
from __future__ import annotations

import numpy as np
import torch

TILE = 128
LATENT = 512


def perm8_col(rho):
    """csrc/common.h: perm8_col -- prepared row rho of an output-side weight holds output column perm8_col(rho)."""
    r = rho & 31
    return (rho & ~31) + 8 * ((r >> 2) & 3) + 4 * (r >> 4) + (r & 3)


def pos_of_unit(u):
    """Column at which hidden unit u is stored in "pos" order (module docstring)."""
    return (u & ~31) + 8 * ((u >> 2) & 3) + 4 * ((u >> 4) & 1) + (u & 3)


def unit_at_pos(n: int = LATENT) -> np.ndarray:
    """unit_at_pos()[p] = the hidden unit stored at column p: rows of a node-term weight are permuted with it (W_pos = W[unit_at_pos])."""
    u = np.arange(n)
    out = np.empty(n, dtype=np.int64)
    out[pos_of_unit(u)] = u
    return out


def pack_segments(recv: np.ndarray, tile: int = TILE) -> np.ndarray:
    """recv: receiver of every edge, non-decreasing.  -> row_edge [n_tiles * tile] int32: the edge id held by each packed row, -1 =
    padding.  Greedy: a receiver's run goes into the current tile if it fits; otherwise the tile is padded and the run starts a new
    tile; a run longer than a tile fills whole tiles and its remainder is treated as a run of its own (continuation piece)."""
    recv = np.asarray(recv)
    n = len(recv)
    if n == 0:
        return np.full(0, -1, dtype=np.int32)
    if np.any(np.diff(recv) < 0):
        raise ValueError("edges must be sorted by receiver")
    starts = np.flatnonzero(np.r_[True, recv[1:] != recv[:-1]])
    lens = np.diff(np.r_[starts, n])
    rows = []
    fill = 0                                              # rows used in the current tile
    for s, ln in zip(starts.tolist(), lens.tolist()):
        while ln > 0:
            room = tile - fill
            if ln <= room:
                rows.append(np.arange(s, s + ln, dtype=np.int32))
                fill = (fill + ln) % tile
                ln = 0
            elif fill == 0:                               # longer than a whole tile: cut
                rows.append(np.arange(s, s + tile, dtype=np.int32))
                s, ln = s + tile, ln - tile
            else:                                         # pad, start a new tile
                rows.append(np.full(room, -1, dtype=np.int32))
                fill = 0
    if fill:
        rows.append(np.full(tile - fill, -1, dtype=np.int32))
    out = np.concatenate(rows)
    assert len(out) % tile == 0
    return out


def continuation_list(row_recv: np.ndarray, tile: int = TILE):
    """Tiles whose first run continues the previous tile's last run (same receiver across the boundary) -- their first partial sum
    goes to the side buffer.  -> (nodes [m], first [m + 1]) CSR over the continuation tiles, grouped by receiver in tile order, and
    the flat tile list: node nodes[i] adds heads[tiles[first[i] : first[i + 1]]] in that order."""
    n_tiles = len(row_recv) // tile
    t = np.arange(1, n_tiles)
    a, b = row_recv[t * tile - 1], row_recv[t * tile]
    cont = t[(a == b) & (b >= 0)]
    if len(cont) == 0:
        return np.zeros(0, np.int32), np.zeros(1, np.int32), np.zeros(0, np.int32)
    nodes_of = row_recv[cont * tile]
    brk = np.flatnonzero(np.r_[True, nodes_of[1:] != nodes_of[:-1]])
    return nodes_of[brk].astype(np.int32), np.r_[brk, len(cont)].astype(np.int32), cont.astype(np.int32)


def to_blocked_f16(x: torch.Tensor) -> torch.Tensor:
    """[R][K] (R % 16 == 0, K % 32 == 0) -> flat fp16 in the blocked layout."""
    R, K = x.shape
    if R % 16 or K % 32:
        raise ValueError("blocked layout needs R % 16 == 0 and K % 32 == 0")
    return x.reshape(R // 16, 16, K // 32, 32).permute(0, 2, 1, 3).to(torch.float16).contiguous().reshape(-1)


def from_blocked_f16(flat: torch.Tensor, R: int, K: int) -> torch.Tensor:
    return flat.reshape(R // 16, K // 32, 16, 32).permute(0, 2, 1, 3).reshape(R, K)


def _planes(v: torch.Tensor, planes: int) -> torch.Tensor:
    """[...] float -> [planes][...] fp16: hi = fp16(v), lo = fp16(v - hi)."""
    v = v.to(torch.float32)
    hi = v.to(torch.float16)
    if planes == 1:
        return hi[None]
    return torch.stack([hi, (v - hi.to(torch.float32)).to(torch.float16)])


_L15 = np.arange(64) & 15
_G = np.arange(64) >> 4
_E = np.arange(8)


def prep_w1_fragments(w1: torch.Tensor, planes: int = 2) -> torch.Tensor:
    """First-Linear weight [H][K] (H % 32 == 0, K % 512 == 0) -> fragment order, flat fp16.  K is taken in SOURCES of 512 columns (the
    node kernel walks concatenated sources one after the other); per source
        block (((j 16 + ks) 2 + n) planes + p),  [lane][e] = plane_p( w1[32 j + 16 n + (lane & 15)][512 s + 32 ks + 8 (lane >> 4) + e] )
    -- one 1 KiB block = the A operand of one v_mfma_f32_16x16x32_f16 (what one ds_read_b128 wave instruction fetches).  A chunk j of 32
    hidden units is contiguous (64 KiB with two planes = one LDS stage); inside it the two 16-unit halves n of a k-step are neighbours,
    so that a step of the kernels' MFMA loops runs four independent accumulator chains."""
    H, K = w1.shape
    if K % 512 or H % 32:
        raise ValueError("prep_w1_fragments: [H][512 n_src] with H % 32 == 0")
    KS = 16
    out = []
    for s in range(K // 512):
        j, ks, n = np.meshgrid(np.arange(H // 32), np.arange(KS), np.arange(2), indexing="ij")
        rows = (32 * j + 16 * n)[..., None, None] + _L15[None, None, None, :, None]            # [J][KS][2][64][1]
        cols = (512 * s + 32 * ks)[..., None, None] + (8 * _G)[None, None, None, :, None] + _E[None, None, None, None, :]
        rows = np.broadcast_to(rows, cols.shape)
        idx = torch.from_numpy((rows * K + cols).reshape(-1)).to(w1.device)
        frag = w1.reshape(-1)[idx].reshape(H // 32, KS, 2, 64, 8)                              # [J][KS][n][lane][e]
        pl = _planes(frag, planes)                                                              # [p][J][KS][n][lane][e]
        out.append(pl.permute(1, 2, 3, 0, 4, 5).contiguous().reshape(-1))
    return torch.cat(out)


def prep_w1_fragments_kouter(w1: torch.Tensor, planes: int = 2) -> torch.Tensor:
    """The same 1 KiB blocks as ``prep_w1_fragments`` in K-OUTER order, for the node kernel (csrc/graphcast_fused.hip: node_mlp_kernel): per
    source, block ((ks 32 + n) planes + p) with n = 2 j + half the 16-unit group -- one LDS stage (64 KiB with two planes, H = 512) = ONE
    k-step of all hidden units."""
    H, K = w1.shape
    if H != 512 or K % 512:
        raise ValueError("prep_w1_fragments_kouter: [512][512 n_src]")
    per = (H // 32) * 16 * 2 * planes * 512                        # elements per source
    flat = prep_w1_fragments(w1, planes)
    out = []
    for s in range(K // 512):
        blk = flat[s * per:(s + 1) * per].reshape(H // 32, 16, 2, planes, 512)     # [j][ks][half][p][lane * 8 + e]
        out.append(blk.permute(1, 0, 2, 3, 4).contiguous().reshape(-1))           # [ks][j][half][p]
    return torch.cat(out)


def prep_w1_node(w1: torch.Tensor) -> torch.Tensor:
    """The node kernel's first Linear, hi/lo planes, in the order ``skgc_node_mlp`` reads (K-outer)."""
    return prep_w1_fragments_kouter(w1)


def prep_w2_fragments(w2: torch.Tensor, planes: int = 2) -> torch.Tensor:
    """Second-Linear weight [N][H] (N % 32 == 0, H % 32 == 0) -> fragment order, flat fp16:
        block (j CF + c) planes + p,  [lane][e] = plane_p( w2[perm8_col(16 c + (lane & 15))][32 j + 16 (e >> 2) + 4 (lane >> 4) + (e & 3)] )
    (k-slot 8 g + e of chunk j = hidden unit 32 j + 16 (e >> 2) + 4 g + (e & 3): the order in which the first Linear's accumulators
    become the second Linear's operand; perm8 rows: a lane's accumulators of a fragment pair are 8 consecutive output columns)."""
    N, H = w2.shape
    CF = N // 16
    j, c = np.meshgrid(np.arange(H // 32), np.arange(CF), indexing="ij")
    rows = perm8_col((16 * c)[..., None, None] + _L15[None, None, :, None])                    # [J][CF][64][1]
    cols = (32 * j)[..., None, None] + (16 * (_E >> 2) + (_E & 3))[None, None, None, :] + (4 * _G)[None, None, :, None]
    rows = np.broadcast_to(rows, cols.shape)
    idx = torch.from_numpy((rows * H + cols).reshape(-1)).to(w2.device)
    frag = w2.reshape(-1)[idx].reshape(H // 32, CF, 64, 8)
    pl = _planes(frag, planes)                                                                  # [p][J][CF][lane][e]
    return pl.permute(1, 2, 0, 3, 4).contiguous().reshape(-1)
