"""CPU pins of the fused GraphCast kernels' contract (csrc/graphcast_fused.hip <-> skyrim_amd/graphcast/fused.py): row packing, blocked
layout, "pos" column order, fragment-order weights -- and a LANE-LEVEL emulation of the kernels' index algebra (MFMA 16x16x32 lane mapping,
hidden-chunk -> k-slot correspondence, perm8 outputs, the segmented DPP scan, the exchange between 16-row groups, continuation pieces)
against plain matrix products.  The arithmetic itself (fp16 planes) is covered by the GPU parity tests; here everything is float64."""
import numpy as np
import pytest
import torch

from skyrim_amd.graphcast import fused as fz

LANES = np.arange(64)
L15, G = LANES & 15, LANES >> 4
L = 512


def test_pack_segments_keeps_runs_whole_and_cuts_only_long_ones():
    rng = np.random.default_rng(0)
    lens = np.r_[rng.integers(1, 43, size=400), 300, rng.integers(1, 43, size=50), 129, 128, 1]
    recv = np.repeat(np.arange(len(lens)) * 3 + 5, lens)
    re = fz.pack_segments(recv)
    assert len(re) % fz.TILE == 0 and np.array_equal(re[re >= 0], np.arange(len(recv)))          # every edge once, order kept
    row_recv = np.where(re >= 0, recv[np.where(re >= 0, re, 0)], -1)
    for t in range(1, len(re) // fz.TILE):
        a, b = row_recv[t * fz.TILE - 1], row_recv[t * fz.TILE]
        if a == b and b >= 0:                                                                        # a run crosses a tile boundary only if it is longer than a tile
            assert lens[(b - 5) // 3] > fz.TILE
    assert (re < 0).sum() < 0.2 * len(re)
    nodes, first, tiles = fz.continuation_list(row_recv)
    assert set(nodes.tolist()) == {5 + 3 * 400, 5 + 3 * 451}                                        # the 300-run and the 129-run
    assert first[-1] == len(tiles) and np.all(row_recv[tiles * fz.TILE] == np.repeat(nodes, np.diff(first)))
    # three edges into every node (mesh -> grid): 126 rows + 2 padding rows per tile, nothing continues
    re3 = fz.pack_segments(np.repeat(np.arange(1000), 3))
    assert np.all((re3.reshape(-1, 128)[:-1, 126:] < 0)) and np.all(re3.reshape(-1, 128)[:-1, :126] >= 0)
    with pytest.raises(ValueError):
        fz.pack_segments(np.array([3, 2, 5]))


def test_blocked_layout_and_pos_order():
    x = torch.arange(32 * 64, dtype=torch.float32).reshape(32, 64) / 8.0
    b = fz.to_blocked_f16(x)
    assert torch.equal(fz.from_blocked_f16(b, 32, 64).float(), x)
    blk_off = lambda row, col, K: (((row >> 4) * (K >> 5) + (col >> 5)) << 9) + ((row & 15) << 5) + (col & 31)     # csrc/common.h  # noqa: E731
    for row, col in ((0, 0), (17, 33), (31, 63), (5, 40)):
        assert b[blk_off(row, col, 64)].item() == x[row, col].item()
    assert np.array_equal(_blocked64(x.double().numpy()), b.double().numpy())                      # the emulation's float64 twin of the layout
    u = np.arange(L)
    pos = fz.pos_of_unit(u)
    assert sorted(pos.tolist()) == u.tolist() and np.array_equal(fz.unit_at_pos()[pos], u)
    # a lane's eight hidden units of chunk j (16 n + 4 g + r) are eight consecutive "pos" columns 32 j + 8 g + [0, 8)
    for j, g in ((0, 0), (3, 2), (15, 3)):
        units = [32 * j + 16 * n + 4 * g + r for n in range(2) for r in range(4)]
        assert fz.pos_of_unit(np.array(units)).tolist() == list(range(32 * j + 8 * g, 32 * j + 8 * g + 8))


# ---- lane-level emulation ---------------------------------------------------------------------------------------------------------- #
def mfma(a, b, acc):
    """v_mfma_f32_16x16x32: a, b [64][8] per-lane operands (A[l & 15][8 (l >> 4) + e], B[8 (l >> 4) + e][l & 15]); acc [64][4] holds D[4 (l >> 4) + r][l & 15]."""
    A = np.zeros((16, 32)); B = np.zeros((32, 16))
    for e in range(8):
        A[L15, 8 * G + e] = a[:, e]
        B[8 * G + e, L15] = b[:, e]
    D = A @ B
    out = acc.copy()
    for r in range(4):
        out[:, r] += D[4 * G + r, L15]
    return out


def frag(flat, block):
    """1 KiB block `block` of a fragment-order tensor as [64 lanes][8]."""
    return flat[block * 512:(block + 1) * 512].reshape(64, 8)


def swish(x):
    return x / (1.0 + np.exp(-x))


def emulate_edge_tile(tile, e_in_b, recv, idx, terms, w1f, w2f, b2, gamma, beta, agg, heads, e_out_b, has_fc1):
    """One workgroup of edge_update_kernel: 4 waves x 2 groups of 16 rows; planes folded (hi + lo) since this is float64."""
    KS, CF, NCH = 16, 32, 16
    tile0 = tile * 128
    y_groups, my_groups = [], []
    for u in range(8):                                             # group u = 2 wave + t
        rows = tile0 + 16 * u + L15                                # per lane
        my = recv[rows]
        rb = (tile0 >> 4) + u
        hh = np.zeros((NCH, 64, 8))
        if has_fc1:
            xh = np.stack([e_in_b[((rb * KS + ks) << 9):((rb * KS + ks + 1) << 9)].reshape(16, 32)[L15][np.arange(64)[:, None], 8 * G[:, None] + np.arange(8)[None, :]]
                           for ks in range(KS)])                   # xh[ks][lane][e] = e[row l15][32 ks + 8 g + e]
        for j in range(NCH):
            hacc = np.zeros((2, 64, 4))
            if not has_fc1:
                st = e_in_b[((rb * KS + j) << 9):((rb * KS + j + 1) << 9)].reshape(16, 32)[L15][np.arange(64)[:, None], 8 * G[:, None] + np.arange(8)[None, :]]
                hacc[0], hacc[1] = st[:, :4], st[:, 4:]
            for s, (tm, ld) in enumerate(terms):
                node = np.where(idx[s][rows] < 0, 0, idx[s][rows])
                base = node * ld + 8 * G + 32 * j
                piece = tm[base[:, None] + np.arange(8)[None, :]]
                hacc[0] += piece[:, :4]; hacc[1] += piece[:, 4:]
            if has_fc1:
                for ks in range(KS):                                # a step = k-step ks against both 16-unit halves n
                    for n in range(2):
                        blk = j * 64 + (ks * 2 + n) * 2             # chunk j = 64 KiB of [ks][n][plane] blocks
                        w = frag(w1f, blk) + frag(w1f, blk + 1)
                        hacc[n] = mfma(w, xh[ks], hacc[n])
            hh[j][:, :4], hh[j][:, 4:] = swish(hacc[0]), swish(hacc[1])
        yacc = np.zeros((CF, 64, 4))
        for j in range(NCH):
            for c in range(CF):
                blk = j * 64 + c * 2
                yacc[c] = mfma(frag(w2f, blk) + frag(w2f, blk + 1), hh[j], yacc[c])
        # LayerNorm: pair bp = columns 32 bp + 8 g + [0, 8)
        y = np.zeros((64, 16, 8))
        for bp in range(16):
            y[:, bp, :4], y[:, bp, 4:] = yacc[2 * bp], yacc[2 * bp + 1]
        col = 32 * np.arange(16)[None, :, None] + 8 * G[:, None, None] + np.arange(8)[None, None, :]
        y = y + b2[col]
        s = y.sum(axis=(1, 2))
        s = s + s[LANES ^ 16]; s = s + s[LANES ^ 32]
        mean = s / L
        q = ((y - mean[:, None, None]) ** 2).sum(axis=(1, 2))
        q = q + q[LANES ^ 16]; q = q + q[LANES ^ 32]
        y = (y - mean[:, None, None]) / np.sqrt(q / L + 1e-5)[:, None, None] * gamma[col] + beta[col]
        if e_out_b is not None:
            for bp in range(16):
                off = ((rb * KS + bp) << 9) + L15 * 32 + G * 8
                for lane in range(64):
                    if my[lane] >= 0:
                        e_out_b[off[lane]:off[lane] + 8] = e_in_b[off[lane]:off[lane] + 8] + y[lane, bp]
        y_groups.append(y); my_groups.append(my)
    # receiver sums: the rows through LDS, one thread per column walks the 128 rows in order (two halves of 256 columns)
    ybuf = np.zeros((128, L))
    for u in range(8):
        for lane in range(64):
            for bp in range(16):
                ybuf[16 * u + L15[lane], 32 * bp + 8 * G[lane]:32 * bp + 8 * G[lane] + 8] = y_groups[u][lane, bp]
    rcv_l = np.r_[recv[tile0 - 1] if tile0 > 0 else -2, recv[tile0:tile0 + 128], -3]
    first = rcv_l[1]
    tile_cont = first >= 0 and rcv_l[0] == first
    acc = np.zeros(L)
    cur = first
    for r in range(128):
        acc = acc + ybuf[r]
        nx = rcv_l[2 + r]
        if nx != cur:
            if cur >= 0:
                (heads[tile] if (tile_cont and cur == first) else agg[cur])[:] = acc
            acc = np.zeros(L)
            cur = nx


def _direct(e, recv_e, send_e, terms_nat, w1e, w2, b2, gamma, beta, n_nodes, static=None):
    pre = (e @ w1e.T) if static is None else static.copy()
    for tm, which in terms_nat:
        pre = pre + tm[send_e if which == "s" else recv_e]
    h = swish(pre)
    z = h @ w2.T + b2
    y = (z - z.mean(1, keepdims=True)) / np.sqrt(z.var(1, keepdims=True) + 1e-5) * gamma + beta
    agg = np.zeros((n_nodes, L))
    np.add.at(agg, recv_e, y)
    return y, agg


@pytest.mark.parametrize("has_fc1", [True, False])
def test_edge_kernel_index_algebra(has_fc1):
    rng = np.random.default_rng(1 if has_fc1 else 2)
    # runs of 1..40 edges plus one of 300 (continues over tiles); receivers are not consecutive integers
    lens = np.r_[rng.integers(1, 41, size=6), 300, rng.integers(1, 20, size=3)]
    n_nodes = 3 * len(lens) + 2
    recv_e = np.repeat(np.arange(len(lens)) * 3 + 1, lens)
    E = len(recv_e)
    send_e = rng.integers(0, n_nodes, size=E)
    row_edge = fz.pack_segments(recv_e)
    R = len(row_edge)
    ok = row_edge >= 0
    re = np.where(ok, row_edge, 0)
    recv = np.where(ok, recv_e[re], -1); send = np.where(ok, send_e[re], -1)
    e = rng.normal(size=(E, L)); w1e = rng.normal(size=(L, L)) / np.sqrt(L); w2 = rng.normal(size=(L, L)) / np.sqrt(L)
    b2, gamma, beta = rng.normal(size=L) * 0.1, 1 + 0.1 * rng.normal(size=L), 0.1 * rng.normal(size=L)
    ts, tr = rng.normal(size=(n_nodes, L)), rng.normal(size=(n_nodes, L))                       # node terms, natural unit order
    upos = fz.unit_at_pos()
    t2 = np.concatenate([ts[:, upos], tr[:, upos]], axis=1).reshape(-1)                            # [nodes][1024] in "pos" order, like the processor's b_ps
    x_packed = np.where(ok[:, None], e[re], 0.0)
    if has_fc1:
        e_b = _blocked64(x_packed)
        y_ref, agg_ref = _direct(e, recv_e, send_e, [(ts, "s"), (tr, "r")], w1e, w2, b2, gamma, beta, n_nodes)
    else:
        e_b = _blocked64(x_packed[:, upos])                                                        # the prepared term in "pos" columns
        y_ref, agg_ref = _direct(e, recv_e, send_e, [(ts, "s"), (tr, "r")], w1e, w2, b2, gamma, beta, n_nodes, static=e)
    w1f = _frag64(fz.prep_w1_fragments, w1e); w2f = _frag64(fz.prep_w2_fragments, w2)
    agg = np.full((n_nodes, L), np.nan); heads = np.zeros((R // 128, L)); e_out = e_b.copy() if has_fc1 else None
    for tile in range(R // 128):
        emulate_edge_tile(tile, e_b, recv, [send, recv], [(t2, 2 * L), (t2[L:], 2 * L)], w1f, w2f, b2, gamma, beta, agg, heads, e_out, has_fc1)
    nodes, first, tiles = fz.continuation_list(recv)
    assert len(nodes) == 1 and first[-1] == 2                                                       # the 300-run: one owner piece + 2 continuation tiles
    for i, nd in enumerate(nodes):
        for k in range(first[i], first[i + 1]):
            agg[nd] += heads[tiles[k]]
    touched = np.unique(recv_e)
    assert np.isnan(np.delete(agg, touched, axis=0)).all()                                          # receivers without rows are not written
    assert np.abs(agg[touched] - agg_ref[touched]).max() < 1e-9
    if has_fc1:
        got = _unblocked64(e_out, R)[ok]
        assert np.abs(got - (e + y_ref)).max() < 1e-9


def _blocked64(x):
    R, K = x.shape
    return x.reshape(R // 16, 16, K // 32, 32).transpose(0, 2, 1, 3).reshape(-1).copy()


def _unblocked64(flat, R, K=L):
    return flat.reshape(R // 16, K // 32, 16, 32).transpose(0, 2, 1, 3).reshape(R, K)


def _frag64(prep, w):
    """The real prep function's INDEX map applied to float64 values: prepare an index-valued matrix, read the permutation back."""
    n = w.size
    assert n < (1 << 24)
    # fp16 cannot hold the indices: recover the permutation exactly with two 11-bit passes
    lo = prep(torch.from_numpy((np.arange(n) % 2048).astype(np.float32)).reshape(w.shape), planes=1).float().numpy().astype(np.int64)
    hi = prep(torch.from_numpy((np.arange(n) // 2048).astype(np.float32)).reshape(w.shape), planes=1).float().numpy().astype(np.int64)
    perm = hi * 2048 + lo
    flat = w.reshape(-1)[perm]                                                                       # planes = 1: [blocks][lane][e]
    # the kernels read (hi, lo) plane pairs: lay the float64 value in the hi block and zeros in the lo block
    out = np.zeros((len(flat) // 512, 2, 512))
    out[:, 0] = flat.reshape(-1, 512)
    return out.reshape(-1)


def test_node_kernel_index_algebra():
    """node_mlp_kernel with two sources: the first Linear K-OUTER on two row groups -- a stage = one k-step of all 512 hidden units in
    [n][plane] order (prep_w1_fragments_kouter), concatenated sources are further k-steps; the accumulators hp[t][half][j] are the hidden
    units 32 j + 16 half + 4 g + r and become the second Linear's k-step-j fragments in place -- then the second Linear chunk by chunk, one
    row group after the other; epilogue rows are fp32 row-major."""
    rng = np.random.default_rng(4)
    x = rng.normal(size=(32, 2 * L)); w1 = rng.normal(size=(L, 2 * L)) / np.sqrt(2 * L); w2 = rng.normal(size=(L, L)) / np.sqrt(L)
    b1 = rng.normal(size=L) * 0.1
    w1k, w2f = _frag64(fz.prep_w1_fragments_kouter, w1), _frag64(fz.prep_w2_fragments, w2)
    hp = np.zeros((2, 2, 16, 64, 4))                               # [t][half][j][lane][r]
    for it in range(32):                                           # (source, k-step): one 64 KiB stage each
        s, ks = divmod(it, 16)
        xk = [x[16 * t + L15][np.arange(64)[:, None], 512 * s + 32 * ks + 8 * G[:, None] + np.arange(8)[None, :]] for t in range(2)]
        for q in range(16):                                        # step q: unit groups 2 q, 2 q + 1 = chunk q, halves 0 / 1
            for half in range(2):
                blk = it * 64 + (q * 2 + half) * 2
                w = frag(w1k, blk) + frag(w1k, blk + 1)
                for t in range(2):
                    hp[t][half][q] = mfma(w, xk[t], hp[t][half][q])
    z = swish(x @ w1.T + b1) @ w2.T
    for t in range(2):
        hh = np.zeros((16, 64, 8))
        for j in range(16):
            hh[j][:, :4] = swish(hp[t][0][j] + b1[32 * j + 4 * G[:, None] + np.arange(4)[None, :]])
            hh[j][:, 4:] = swish(hp[t][1][j] + b1[32 * j + 16 + 4 * G[:, None] + np.arange(4)[None, :]])
        yacc = np.zeros((32, 64, 4))
        for j in range(16):
            for q in range(8):
                for i in range(4):
                    blk = j * 64 + q * 8 + i * 2
                    yacc[4 * q + i] = mfma(frag(w2f, blk) + frag(w2f, blk + 1), hh[j], yacc[4 * q + i])
        got = np.zeros((16, L))
        for bp in range(16):
            for i in range(8):
                got[L15, 32 * bp + 8 * G + i] = (yacc[2 * bp] if i < 4 else yacc[2 * bp + 1])[:, i & 3]
        assert np.abs(got - z[16 * t:16 * t + 16]).max() < 1e-9, t
    # the K-outer order is a permutation of the chunk order's blocks
    a, b = fz.prep_w1_fragments(torch.from_numpy(w1).float()), fz.prep_w1_fragments_kouter(torch.from_numpy(w1).float())
    assert a.numel() == b.numel() and torch.equal(torch.sort(a.reshape(-1, 512).float().sum(1))[0], torch.sort(b.reshape(-1, 512).float().sum(1))[0])


def test_prep_w1_node_is_the_k_outer_order():
    """fused.prep_w1_node: what skgc_node_mlp reads -- a permutation of the chunk-order 1 KiB blocks."""
    w = torch.randn(512, 1024, generator=torch.Generator().manual_seed(0))
    got = fz.prep_w1_node(w)
    assert torch.equal(got, fz.prep_w1_fragments_kouter(w)) and not torch.equal(got, fz.prep_w1_fragments(w))
    key = lambda t: sorted(map(bytes, t.reshape(-1, 512).numpy().view(np.uint8)))  # noqa: E731
    assert key(got) == key(fz.prep_w1_fragments(w))
