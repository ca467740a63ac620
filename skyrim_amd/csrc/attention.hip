// Earth-specific 3D window attention (EarthAttention3D core) for gfx950.
//
// One wavefront = one (window, head): 144 tokens x head_dim 32.  Everything a wave needs is read
// straight into MFMA fragment registers -- Q/K rows are 64-byte contiguous, V arrives transposed
// ([d][token], written that way by the QKV epilogue) so each lane's 4 keys are 8 contiguous bytes,
// and the earth-specific bias (+ shifted-window mask) is pre-expanded per (window type, head) in
// exactly the accumulator layout.  No LDS, no barriers; the only cross-lane traffic is the 2-step
// butterfly of the softmax row reduction.
//
//   S^T[key][q] = K Q^T   (A = K frag, B = Q frag)   -> lane (q = l&15) holds 4 keys per fragment
//   O^T[d][q]   = V^T P^T (A = V^T frag, B = P frag) -> lane (q = l&15) holds 4 head-dim rows per fragment
//
// Which K row / V^T row sits in which fragment row is free (it is only an address), and is chosen so that every
// HBM/L2 access of the kernel is 16 bytes per lane (8-byte accesses cost the same per wave instruction and therefore
// twice per byte; the CU's memory pipeline is what bounds this kernel -- DESIGN.md 5.2):
//   * key blocks kb = 0..3: row 4g+i of K fragment (2kb + jj) is key 32kb + 8g + 4jj + i, so a lane's accumulators of the
//     fragment pair are the 8 CONSECUTIVE keys 32kb + 8g + [0..7] = exactly MFMA k-slots 8g..8g+7 of the PV product:
//     the softmax output feeds the second MFMA without data movement, the V^T fragment is one 16-byte load of 8
//     consecutive tokens, and the bias fragment pair is one 16-byte load; fragment 8 (keys 128..143) keeps the plain
//     order 128 + 4g + i with 8-byte accesses;
//   * row 4g+i of V^T fragment df is head-dim row 8g + 4df + i, so a lane's two output accumulators are 8 consecutive
//     columns of the attention output: one 16-byte store per plane.
#include <cstdlib>
#include "gemm_dma.h"
#include "launchers.h"

namespace skp {

// key held by fragment row `row` (= 4g + i) of K fragment f (see the header comment)
__device__ __forceinline__ int attn_key(int f, int row) {
    return f < 8 ? 32 * (f >> 1) + 8 * (row >> 2) + 4 * (f & 1) + (row & 3) : 128 + row;
}

// Softmax over the 144 keys of a query block + P V, on scores that already carry bias and mask.  The kernels are bound by VALU issue,
// not by bytes (PMC: 31 % issuing, 56 % issue-stalled, matrix pipes 11 % busy), so the VALU work per score is what counts:
//   * v_exp_f32 directly (__builtin_amdgcn_exp2f): exp2f() wraps it in a denormal-range rescue (compare, two selects, add, ldexp: six
//     instructions per value); an argument below -126 is a softmax weight below 1e-38 and flushes to zero either way;
//   * the row sum comes out of the matrix pipe: one more V^T "fragment" of ones gives every lane the sum over ALL keys of its query --
//     five MFMAs on an idle pipe instead of 36 adds and two shuffles, and it sums the fp16-rounded P that multiplies V;
//   * 1 / sum by v_rcp_f32 (1 ulp; the division expands to a ten-instruction sequence).
__device__ __forceinline__ void attn_softmax_pv(f32x4 (&s)[9], const uint4 (&vf)[2][5], f32x4 (&o)[2], f32x4& osum) {
    typedef f16 T;
    const float LOG2E = 1.4426950408889634f;
    float mx = s[0][0];
#pragma unroll
    for (int f = 0; f < 9; ++f)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[f][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 16));
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    const float mxl = mx * LOG2E;
#pragma unroll
    for (int f = 0; f < 9; ++f)
#pragma unroll
        for (int r = 0; r < 4; ++r) s[f][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[f][r], LOG2E, -mxl));
    o[0] = f32x4{0.f, 0.f, 0.f, 0.f}; o[1] = f32x4{0.f, 0.f, 0.f, 0.f}; osum = f32x4{0.f, 0.f, 0.f, 0.f};
    const uint4 ones = make_uint4(0x3C003C00u, 0x3C003C00u, 0x3C003C00u, 0x3C003C00u);      // 8 x fp16 1.0
#pragma unroll
    for (int kb = 0; kb < 5; ++kb) {
        float pv[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            pv[r] = s[2 * kb][r];
            pv[4 + r] = (kb < 4) ? s[(2 * kb + 1) % 9][r] : 0.f;
        }
        uint4 pf[1];
        split8<T, 1>(pv, pf);
#pragma unroll
        for (int df = 0; df < 2; ++df) o[df] = OpT<T>::mfma(as_v8<T>(vf[df][kb]), as_v8<T>(pf[0]), o[df]);
        osum = OpT<T>::mfma(as_v8<T>(ones), as_v8<T>(pf[0]), osum);
    }
}

// Q, K, V and the softmax output P are single fp16 planes in EVERY precision mode: attention is the least
// rounding-sensitive part of the network (measured on the oracle: fp16 q/k/v -> 7e-5, fp16 P -> 4e-5 per-channel
// error, vs 3.3e-4 / 3.6e-4 for the attention output / MLP hidden, which therefore stay hi/lo split), so
// QK^T and PV are one MFMA term each and the Q/K/V round trip through HBM is 2 bytes per element.
// 165 VGPRs -> three waves per SIMD (the default schedule takes 189 = two; four would spill ~40 registers)
template <class TO, int NPL_O>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) earth_attention_kernel(const f16* __restrict__ q, const f16* __restrict__ k,
                                                              const f16* __restrict__ vt, long long plane,
                                                              const f16* __restrict__ bias_exp, TO* __restrict__ out, long long out_plane,
                                                              int ld_out, int n_win, int nW, int heads) {
    typedef f16 T;
    constexpr int NPL = 1;
    const int lane = threadIdx.x & 63;
    const long long wg = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wg >= (long long)n_win * heads) return;
    // consecutive waves share (type, head) -> the expanded bias tile stays hot in L2
    const int wi = (int)(wg % nW);
    const int th = (int)(wg / nW);
    const int head = th % heads, type = th / heads;
    const int win = type * nW + wi;
    const long long base = (long long)win * heads + head;
    const int l15 = lane & 15, g = lane >> 4;

    const T* qp = q + base * (WIN_TOKENS * HEAD_DIM) + l15 * HEAD_DIM + g * 8;
    const T* kp = k + base * (WIN_TOKENS * HEAD_DIM) + g * 8;
    const T* vp = vt + base * (WIN_TOKENS * HEAD_DIM) + (8 * (l15 >> 2) + (l15 & 3)) * WIN_TOKENS;      // + 4 df rows
    const f16* bp = bias_exp + ((long long)type * heads + head) * (81 * 256);

    uint4 kf[NPL][9];
#pragma unroll
    for (int p = 0; p < NPL; ++p)
#pragma unroll
        for (int f = 0; f < 9; ++f) kf[p][f] = *reinterpret_cast<const uint4*>(kp + p * plane + attn_key(f, l15) * HEAD_DIM);

    uint4 vf[NPL][2][5];
#pragma unroll
    for (int p = 0; p < NPL; ++p)
#pragma unroll
        for (int df = 0; df < 2; ++df)
#pragma unroll
            for (int kb = 0; kb < 5; ++kb) {
                const T* s = vp + p * plane + df * 4 * WIN_TOKENS;
                if (kb < 4) vf[p][df][kb] = *reinterpret_cast<const uint4*>(s + kb * 32 + g * 8);
                else { const uint2 lo = *reinterpret_cast<const uint2*>(s + 128 + g * 4); vf[p][df][kb] = make_uint4(lo.x, lo.y, 0, 0); }
            }

    // software prefetch: Q fragment and the 9 bias/mask fragments of query block qf+1 are in flight while block qf
    // computes (one exposed L2/HBM round trip per window instead of nine)
    uint4 qn[NPL];
    uint4 bn[4];
    uint2 bn8;
#pragma unroll
    for (int p = 0; p < NPL; ++p) qn[p] = *reinterpret_cast<const uint4*>(qp + p * plane);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) bn[kb] = *reinterpret_cast<const uint4*>(bp + kb * 512 + lane * 8);
    bn8 = *reinterpret_cast<const uint2*>(bp + 2048 + lane * 4);
#pragma unroll 1
    for (int qf = 0; qf < 9; ++qf) {
        uint4 qv[NPL];
        uint4 bcur[4];
#pragma unroll
        for (int p = 0; p < NPL; ++p) qv[p] = qn[p];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) bcur[kb] = bn[kb];
        const uint2 bcur8 = bn8;
        if (qf < 8) {
#pragma unroll
            for (int p = 0; p < NPL; ++p) qn[p] = *reinterpret_cast<const uint4*>(qp + p * plane + (qf + 1) * 16 * HEAD_DIM);
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) bn[kb] = *reinterpret_cast<const uint4*>(bp + (qf + 1) * 2304 + kb * 512 + lane * 8);
            bn8 = *reinterpret_cast<const uint2*>(bp + (qf + 1) * 2304 + 2048 + lane * 4);
        }

        // the bias / mask tile is the accumulator the score MFMA starts from (one conversion per value, no add)
        f32x4 s[9];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const typename OpT<f16>::v8 b = as_v8<f16>(bcur[kb]);
            s[2 * kb] = f32x4{(float)b[0], (float)b[1], (float)b[2], (float)b[3]};
            s[2 * kb + 1] = f32x4{(float)b[4], (float)b[5], (float)b[6], (float)b[7]};
        }
        {
            typedef f16 h4 __attribute__((ext_vector_type(4)));
            const h4 b = __builtin_bit_cast(h4, bcur8);
            s[8] = f32x4{(float)b[0], (float)b[1], (float)b[2], (float)b[3]};
        }
#pragma unroll
        for (int f = 0; f < 9; ++f) s[f] = OpT<T>::mfma(as_v8<T>(kf[0][f]), as_v8<T>(qv[0]), s[f]);
        f32x4 o[2], osum;
        attn_softmax_pv(s, vf[0], o, osum);
        const float inv = __builtin_amdgcn_rcpf(osum[0]);
        // blocked [row/16][col/32][16][32] layout: this head's 32 columns are exactly one column block
        TO* orow = out + blk_off((long long)win * WIN_TOKENS + qf * 16 + l15, head * HEAD_DIM, ld_out) + g * 8;
        const float y[8] = {o[0][0] * inv, o[0][1] * inv, o[0][2] * inv, o[0][3] * inv, o[1][0] * inv, o[1][1] * inv, o[1][2] * inv, o[1][3] * inv};
        store8_planes<TO, NPL_O>(orow, out_plane, y);
    }
}


// ------------------------------------------------------------------------------------------------------------------------------- //
//  The same attention with the earth-specific bias GATHERED from its compact table ("bias gathered from the compact
//  (3312, type, head) table in-kernel"): the expanded tiles above are 41 KB of the 86 KB a wave moves, and the kernel is bound by the
//  CU's memory pipeline.  One workgroup = one (window type, head); its four waves walk the type's nW longitude windows, which all
//  share ONE bias.  The workgroup stages that bias in LDS once, from the prepared [144][24] fp16 form (prep_bias_compact: row =
//  (z_q + 2 z_k) 36 + (h_q + 6 h_k), column = w_k - w_q + 11, shifted-window mask folded into the rows): a lane's four consecutive keys
//  are then four consecutive fp16 of one row -- one ds_read_b64, IF the address is 8-byte aligned, which depends on w_q mod 4 only:
//  the table is stored four times, copy c shifted by c entries, and a lane reads copy (w_q + 1) mod 4 (32 KB of LDS, 81 reads per
//  window and lane).  Per window a wave now moves 27 KB of Q / K / V and 18 KB of output.
// fp16 entries per table row / per shifted copy.  The 120-entry pad between the copies spreads the four copies (which a 16-query fragment mixes:
// copy = (w_q + 1) mod 4) over the LDS banks: simulated over every (query fragment, key group) read of the kernel, a ds_read_b64 half-wave
// costs 1.99 LDS cycles on average (worst 2) instead of 3.13 (worst 4) with the copies back to back -- round 3's counters showed 57 % of the
// kernel's LDS-active cycles as bank conflicts.
constexpr int BT_ROW = 28, BT_COPY = 144 * BT_ROW + 120;

template <class TO, int NPL_O>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) earth_attention2_kernel(const f16* __restrict__ q, const f16* __restrict__ k,
                                                              const f16* __restrict__ vt, const f16* __restrict__ bias_cmp, TO* __restrict__ out,
                                                              long long out_plane, int ld_out, int nW, int heads) {
    typedef f16 T;
    __shared__ __attribute__((aligned(16))) f16 tabs[4 * BT_COPY];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int th = blockIdx.x;
    const int head = th % heads, type = th / heads;
    const int l15 = lane & 15, g = lane >> 4;
    {
        const f16* src = bias_cmp + (long long)th * 3456;
        for (int i = threadIdx.x; i < 3456; i += 256) {
            const int r = i / 24, e = i - 24 * r;
            const f16 v = src[i];
            if (e < 23) {
#pragma unroll
                for (int c = 0; c < 4; ++c) tabs[c * BT_COPY + r * BT_ROW + e + c] = v;
            }
        }
    }
    // byte offset of a lane's key group (4 consecutive keys k0 ..): rows of (z_k, h_k), column w_k0
    int koff[9];
#pragma unroll
    for (int f = 0; f < 9; ++f) {
        const int k0 = f < 8 ? 32 * (f >> 1) + 8 * g + 4 * (f & 1) : 128 + 4 * g;
        const int zk = k0 / 72, hk = (k0 / 12) % 6, wk0 = k0 % 12;
        koff[f] = 2 * ((2 * zk * 36 + 6 * hk) * BT_ROW + wk0);
    }
    __syncthreads();
    const char* tb = reinterpret_cast<const char*>(tabs);

    for (int wi = wave; wi < nW; wi += 4) {
        const int win = type * nW + wi;
        const long long base = (long long)win * heads + head;
        const T* qp = q + base * (WIN_TOKENS * HEAD_DIM) + l15 * HEAD_DIM + g * 8;
        const T* kp = k + base * (WIN_TOKENS * HEAD_DIM) + g * 8;
        const T* vp = vt + base * (WIN_TOKENS * HEAD_DIM) + (8 * (l15 >> 2) + (l15 & 3)) * WIN_TOKENS;      // + 4 df rows

        uint4 kf[9];
#pragma unroll
        for (int f = 0; f < 9; ++f) kf[f] = *reinterpret_cast<const uint4*>(kp + attn_key(f, l15) * HEAD_DIM);
        uint4 vf[2][5];
#pragma unroll
        for (int df = 0; df < 2; ++df)
#pragma unroll
            for (int kb = 0; kb < 5; ++kb) {
                const T* s = vp + df * 4 * WIN_TOKENS;
                if (kb < 4) vf[df][kb] = *reinterpret_cast<const uint4*>(s + kb * 32 + g * 8);
                else { const uint2 lo = *reinterpret_cast<const uint2*>(s + 128 + g * 4); vf[df][kb] = make_uint4(lo.x, lo.y, 0, 0); }
            }
        uint4 qn = *reinterpret_cast<const uint4*>(qp);
#pragma unroll 1
        for (int qf = 0; qf < 9; ++qf) {
            const uint4 qv = qn;
            if (qf < 8) qn = *reinterpret_cast<const uint4*>(qp + (qf + 1) * 16 * HEAD_DIM);
            // the lane's query: q = 16 qf + l15 -> (z_q, h_q, w_q); copy c = (w_q + 1) mod 4 makes every group address a multiple of 8 bytes
            const int qi = 16 * qf + l15;
            const int zq = qi >= 72 ? 1 : 0, hq = (qi - 72 * zq) / 12, wq = qi - 72 * zq - 12 * hq;
            const int c = (wq + 1) & 3;
            const char* bq = tb + 2 * (c * BT_COPY + (zq * 36 + hq) * BT_ROW + (11 - wq) + c);
            uint2 bias[9];
#pragma unroll
            for (int f = 0; f < 9; ++f) bias[f] = *reinterpret_cast<const uint2*>(bq + koff[f]);

            f32x4 s[9];
#pragma unroll
            for (int f = 0; f < 9; ++f) {
                typedef f16 h4 __attribute__((ext_vector_type(4)));
                const h4 b = __builtin_bit_cast(h4, bias[f]);
                s[f] = OpT<T>::mfma(as_v8<T>(kf[f]), as_v8<T>(qv), f32x4{(float)b[0], (float)b[1], (float)b[2], (float)b[3]});
            }
            f32x4 o[2], osum;
            attn_softmax_pv(s, vf, o, osum);
            const float inv = __builtin_amdgcn_rcpf(osum[0]);
            TO* orow = out + blk_off((long long)win * WIN_TOKENS + qf * 16 + l15, head * HEAD_DIM, ld_out) + g * 8;
            const float y[8] = {o[0][0] * inv, o[0][1] * inv, o[0][2] * inv, o[0][3] * inv, o[1][0] * inv, o[1][1] * inv, o[1][2] * inv, o[1][3] * inv};
            store8_planes<TO, NPL_O>(orow, out_plane, y);
        }
    }
}


// ------------------------------------------------------------------------------------------------------------------------------- //
//  QKV + attention as ONE kernel (round 6): q / k / v never reach HBM.
//
//  rounds 2-5 ran the QKV Linear (rt_qkv_kernel: stream hi plane -> Q, K, V^T, 3 x 2 bytes per window token and channel written) and the
//  attention above (the same bytes read back) as two launches -- 5.3 of the step's 16.3 ms and about half of its HBM traffic.  Here a
//  WAVE owns one (window, head) for both: it computes that head's 144 x 32 Q, K and V from the window's 144 stream rows, keeps them in
//  registers in exactly the fragment forms the attention loop consumes, and goes on as earth_attention2_kernel does.
//
//  Why no data movement is needed between the two halves (16x16x32 MFMA: an A fragment is lane (row l&15, k-group l>>4), a B fragment is lane
//  (column l&15, k-group l>>4) -- the SAME per-lane bytes of a [16 tokens][32 channels] tile serve as either):
//    * one stream-row fragment xf of 16 window tokens (token attn_key(f, l&15): WHICH token sits in which fragment row is only an address)
//      is the B operand of  D^T = W_q xf^T  and  D^T = W_k xf^T  (a lane ends up with dims 8g + [0..7] of its token after the perm8 row order of
//      the prepared weights: the B operand Q^T and the A operand K of  S^T = K Q^T) and the A operand of  D = xf W_v^T  (a lane ends up with 4
//      consecutive keys of one head-dim row; two token fragments = the 8 keys 32 kb + 8 g + [0..7] of a V^T fragment of  O^T = V^T P^T);
//    * the weights are the 32-column blocks prep_rowtile_weights already writes (block j = q head j, heads + j = k, 2 heads + j = v; [ks][n]
//      [plane] KiB each): the head's three blocks are fetched into LDS once per workgroup by LDS-DMA and stay there for every window the
//      workgroup's waves walk (one workgroup = one (window type, head), as above: one bias table, one set of weights);
//    * token fragments are processed THREE at a time, so a weight fragment read from LDS feeds three MFMAs per plane.
//  LDS: 72 KiB of weights (C = 384 with one plane, C = 192 with hi / lo planes) + the 32 KiB bias copies: one 8-wave workgroup per CU, two
//  waves per SIMD.  Queries are walked in the key order too (query fragment qf = tokens attn_key(qf, .)): only the bias row and the output
//  row of a lane change.
template <int C_, int PL_>
struct QaShape {
    static constexpr int C = C_, PL = PL_, KS = C / 32, NWAVES = 8, THREADS = 64 * NWAVES;
    static constexpr int BLK_KIB = KS * 2 * PL;                 // one 32-column weight block: [ks][n][plane] KiB
    static constexpr int W_BYTES = 3 * BLK_KIB * 1024;          // the head's q, k and v blocks
    static constexpr int TAB_BYTES = (4 * BT_COPY * 2 + 1023) / 1024 * 1024;
    static constexpr int RING_BYTES = 6 * 1024;                 // per wave: two k-steps x three 1 KiB stream-row fragments
    static constexpr int SMEM = W_BYTES + TAB_BYTES + 1024 + NWAVES * RING_BYTES;
    static_assert(SMEM <= 160 * 1024, "LDS");
};

struct QkvAttnArgs {
    const f16* xs;                 // residual stream, hi plane, blocked layout
    const int* widx;               // window row -> stream token, -1 = padding
    const f16* wf;                 // qkv weights [3C][C] in fragment order (prep_rowtile_weights), PL planes
    const float* bias;             // [3C]
    const f16* bias_cmp;           // compact earth-specific bias tables (+ mask), [type][head][144][24]
    const f16* zrow;               // zeros: the row of a padding token
    f16* out; long long out_plane; // attention output, window-ordered rows, blocked layout
    int nW, heads, types;
    float scale;
};

template <class S, int NPL_O>
__global__ void __launch_bounds__(S::THREADS) __attribute__((amdgpu_waves_per_eu(2, 2)))
qkv_attention_kernel(const QkvAttnArgs a) {
    typedef f16 T;
    constexpr int C = S::C, KS = S::KS, PL = S::PL;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* wl = smem;
    f16* tabs = reinterpret_cast<f16*>(smem + S::W_BYTES);
    float* qb = reinterpret_cast<float*>(smem + S::W_BYTES + S::TAB_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Workgroup -> (window type, head), XCD-aware.  The `heads` workgroups of a type read the SAME stream rows (the type's nW windows); workgroups
    // go to the 8 XCDs round-robin in launch order, each XCD with its own L2.  With the plain map (type major, head minor) the heads of a type
    // landed on different XCDs and every L2 fetched the rows for itself: 0.94 GB fetched per launch at C = 384 for a 0.1 GB operand, the kernel
    // running at 5.4 TB/s of fabric traffic (profiles/r06_pangu_pmc.json, first take).  Here XCD x owns the types x, x + 8, ...: block b -> XCD
    // b & 7, slot b >> 3 = (type index on that XCD) * heads + head, so a type's heads run side by side on ONE L2.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int head = slot % a.heads, type = xcd + 8 * (slot / a.heads);
    if (type >= a.types) return;                                 // (the XCDs with one type fewer: whole workgroups, before any barrier)
    const int th = type * a.heads + head;
    {   // the head's three weight blocks -> LDS, piece p of a block by wave p % 8
        const unsigned lds_base = (unsigned)(size_t)smem;
#pragma unroll
        for (int which = 0; which < 3; ++which) {
            const f16* src = a.wf + ((long long)(which * a.heads + head) * S::BLK_KIB << 9) + lane * 8;
            for (int p = wave; p < S::BLK_KIB; p += S::NWAVES) glds16(src + (p << 9), lds_base + (unsigned)((which * S::BLK_KIB + p) << 10));
        }
        // the compact bias table -> its four shifted LDS copies: all of a thread's entries are requested first, then written (the loop form
        // of earth_attention2_kernel pays one L2 round trip per entry: seven in a row here, per workgroup, with nothing else resident on the CU)
        const f16* src = a.bias_cmp + (long long)th * 3456;
        constexpr int NT = (3456 + S::THREADS - 1) / S::THREADS;
        f16 tv[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) { const int i = tid + j * S::THREADS; tv[j] = src[i < 3456 ? i : 0]; }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int i = tid + j * S::THREADS;
            const int r = i / 24, e = i - 24 * r;
            if (i < 3456 && e < 23) {
#pragma unroll
                for (int c = 0; c < 4; ++c) tabs[c * BT_COPY + r * BT_ROW + e + c] = tv[j];
            }
        }
        if (tid < 96) qb[tid] = a.bias[(tid >> 5) * C + head * HEAD_DIM + (tid & 31)];
    }
    int koff[9];
#pragma unroll
    for (int f = 0; f < 9; ++f) {
        const int k0 = f < 8 ? 32 * (f >> 1) + 8 * g + 4 * (f & 1) : 128 + 4 * g;
        const int zk = k0 / 72, hk = (k0 / 12) % 6, wk0 = k0 % 12;
        koff[f] = 2 * ((2 * zk * 36 + 6 * hk) * BT_ROW + wk0);
    }
    // the window table entries of a window's nine token fragments (9 dependent 4-byte gathers) are fetched ONE WINDOW AHEAD, under the attention
    // loop (the first window's: under the weight fetch); a token-group triple's first two ring fetches are issued before the previous triple's
    // conversions: what a triple waits for at its top is then one L2 round trip that has been in flight for a while, not three in a row
    int srcn[9];
    auto load_src = [&](int wi_) {
        const int w_ = type * a.nW + wi_;
#pragma unroll
        for (int f = 0; f < 9; ++f) srcn[f] = a.widx[w_ * WIN_TOKENS + attn_key(f, l15)];
    };
    load_src(wave < a.nW ? wave : 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const char* tb = reinterpret_cast<const char*>(tabs);
    const char* wrd = wl + lane * 16;
    const unsigned ring_base = (unsigned)(size_t)smem + (unsigned)(S::W_BYTES + S::TAB_BYTES + 1024 + wave * S::RING_BYTES);
    const char* ring_rd = smem + S::W_BYTES + S::TAB_BYTES + 1024 + wave * S::RING_BYTES + lane * 16;
    // (a lane's share of the biases -- q / k dims 8g + [0..7]; v: the dim of column l15 in fragment df -- is read from LDS where it is used:
    // 18 registers held across the loops were 18 registers spilled)

    for (int wi = wave; wi < a.nW; wi += S::NWAVES) {
        const int win = type * a.nW + wi;
        int src[9];
#pragma unroll
        for (int f = 0; f < 9; ++f) src[f] = srcn[f];
        uint4 qv[9], kf[9];
        uint2 vh[9][2];
        const T* xp[3];
        int xstep[3];
        auto point = [&](int tg_) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int r = src[3 * tg_ + i];
                xp[i] = r >= 0 ? a.xs + blk_off(r, g * 8, C) : a.zrow;
                xstep[i] = r >= 0 ? 512 : 0;
            }
        };
        auto fetch = [&](int ks) {
#pragma unroll
            for (int i = 0; i < 3; ++i) glds16(xp[i] + ks * xstep[i], ring_base + (unsigned)(((ks & 1) * 3 + i) << 10));
        };
        point(0);
        fetch(0);
        fetch(1);
#pragma unroll
        for (int tg = 0; tg < 3; ++tg) {
            f32x4 qa[3][2], ka[3][2], va[3][2];
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int n = 0; n < 2; ++n) { qa[i][n] = f32x4{0.f, 0.f, 0.f, 0.f}; ka[i][n] = f32x4{0.f, 0.f, 0.f, 0.f}; va[i][n] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            // The stream rows reach the wave through its own two-slot LDS ring, fetched by LDS-DMA two k-steps ahead (a gather: every lane hands its
            // own global address, the fragment lands lane-linear): a k-step is 18 MFMAs -- ~300 clocks -- and an L2 round trip under load is several
            // times that, so with register prefetch one k-step ahead (the first form of this kernel) every k-step began with a wait: 0.38 ms per
            // launch where the two-launch form took 0.29.  Only the issuing wave reads its ring: s_waitcnt vmcnt, no barrier.
#pragma unroll 1
            for (int ks = 0; ks < KS; ++ks) {
                if (ks + 1 < KS) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");        // the three requests of k-step ks + 1 may still be in flight
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                uint4 xf[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) xf[i] = *reinterpret_cast<const uint4*>(ring_rd + (((ks & 1) * 3 + i) << 10));
                if (ks + 2 < KS) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                     // the slot is in registers before it is overwritten
                    fetch(ks + 2);
                }
#pragma unroll
                for (int which = 0; which < 3; ++which)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
#pragma unroll
                        for (int p = PL - 1; p >= 0; --p) {                    // the lo plane first, the hi plane last
                            const uint4 w = *reinterpret_cast<const uint4*>(wrd + ((((which * KS + ks) * 2 + n) * PL + p) << 10));
#pragma unroll
                            for (int i = 0; i < 3; ++i) {
                                if (which == 0) qa[i][n] = OpT<T>::mfma(as_v8<T>(w), as_v8<T>(xf[i]), qa[i][n]);
                                else if (which == 1) ka[i][n] = OpT<T>::mfma(as_v8<T>(w), as_v8<T>(xf[i]), ka[i][n]);
                                else va[i][n] = OpT<T>::mfma(as_v8<T>(xf[i]), as_v8<T>(w), va[i][n]);
                            }
                        }
            }
            if (tg < 2) {                                   // the next triple's first two k-steps: both slots are free (their last reads fed the MFMAs above)
                point(tg + 1);
                fetch(0);
                fetch(1);
            }
            float bq[8], bk[8], bv[2];
#pragma unroll
            for (int i = 0; i < 8; ++i) { bq[i] = qb[8 * g + i]; bk[i] = qb[32 + 8 * g + i]; }
#pragma unroll
            for (int df = 0; df < 2; ++df) bv[df] = qb[64 + 8 * (l15 >> 2) + 4 * df + (l15 & 3)];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int f = 3 * tg + i;
                float v[8];
                uint4 o[1];
#pragma unroll
                for (int r = 0; r < 4; ++r) { v[r] = (qa[i][0][r] + bq[r]) * a.scale; v[4 + r] = (qa[i][1][r] + bq[4 + r]) * a.scale; }
                split8<T, 1>(v, o);
                qv[f] = o[0];
#pragma unroll
                for (int r = 0; r < 4; ++r) { v[r] = ka[i][0][r] + bk[r]; v[4 + r] = ka[i][1][r] + bk[4 + r]; }
                split8<T, 1>(v, o);
                kf[f] = o[0];
#pragma unroll
                for (int df = 0; df < 2; ++df) {
                    const float u[4] = {va[i][df][0] + bv[df], va[i][df][1] + bv[df], va[i][df][2] + bv[df], va[i][df][3] + bv[df]};
                    uint2 h[1];
                    split4<T, 1>(u, h);
                    vh[f][df] = h[0];
                }
            }
        }
        uint4 vf[2][5];
#pragma unroll
        for (int df = 0; df < 2; ++df) {
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) vf[df][kb] = make_uint4(vh[2 * kb][df].x, vh[2 * kb][df].y, vh[2 * kb + 1][df].x, vh[2 * kb + 1][df].y);
            vf[df][4] = make_uint4(vh[8][df].x, vh[8][df].y, 0, 0);
        }
        if (wi + S::NWAVES < a.nW) load_src(wi + S::NWAVES);
        // A ROLLED loop over the query fragments: qv[qf] is then indexed at run time and lives in scratch memory (144 bytes per lane, written once
        // and read once per window, L1-resident).  Fully unrolled the nine softmax bodies were scheduled into each other: 78-105 spilled registers
        // and 0.28 ms per launch at C = 384 against 0.235 in this form (measured; the two-launch form: 0.29)
#pragma unroll 1
        for (int qf = 0; qf < 9; ++qf) {
            // the lane's query: the token of fragment row l15 -> (z_q, h_q, w_q); copy c = (w_q + 1) mod 4 makes every group address a multiple of 8 bytes
            const int qi = attn_key(qf, l15);
            const int zq = qi >= 72 ? 1 : 0, hq = (qi - 72 * zq) / 12, wq = qi - 72 * zq - 12 * hq;
            const int c = (wq + 1) & 3;
            const char* bqp = tb + 2 * (c * BT_COPY + (zq * 36 + hq) * BT_ROW + (11 - wq) + c);
            f32x4 s[9];
#pragma unroll
            for (int f = 0; f < 9; ++f) {
                typedef f16 h4 __attribute__((ext_vector_type(4)));
                const h4 b = __builtin_bit_cast(h4, *reinterpret_cast<const uint2*>(bqp + koff[f]));
                s[f] = OpT<T>::mfma(as_v8<T>(kf[f]), as_v8<T>(qv[qf]), f32x4{(float)b[0], (float)b[1], (float)b[2], (float)b[3]});
            }
            f32x4 o[2], osum;
            attn_softmax_pv(s, vf, o, osum);
            const float inv = __builtin_amdgcn_rcpf(osum[0]);
            T* orow = a.out + blk_off((long long)win * WIN_TOKENS + qi, head * HEAD_DIM, C) + g * 8;
            const float y[8] = {o[0][0] * inv, o[0][1] * inv, o[0][2] * inv, o[0][3] * inv, o[1][0] * inv, o[1][1] * inv, o[1][2] * inv, o[1][3] * inv};
            store8_planes<T, NPL_O>(orow, a.out_plane, y);
        }
    }
}

template <class S>
static hipError_t launch_qa(const QkvAttnArgs& a, int types, int out_planes, hipStream_t stream) {
    if (out_planes == 1) {
        auto kern = qkv_attention_kernel<S, 1>;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, S::SMEM);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3((unsigned)(8 * ((types + 7) / 8) * a.heads)), dim3(S::THREADS), S::SMEM, stream, a);
    } else {
        auto kern = qkv_attention_kernel<S, 2>;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, S::SMEM);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3((unsigned)(8 * ((types + 7) / 8) * a.heads)), dim3(S::THREADS), S::SMEM, stream, a);
    }
    return hipGetLastError();
}

// QKV (from the stream's hi plane: one term against the one-plane weights b.qkvh, or two against hi / lo b.qkvf) + window attention in one launch
hipError_t op_qkv_attention(const Geom& g, const BlockW<f16>& b, const int* widx, int res, const f16* Xs, const Work<PrecF16x3>& wk, hipStream_t s, int out_planes) {
    if (!b.bias_cmp || !(b.qkvh || b.qkvf)) return hipErrorInvalidValue;
    const int heads = res == 0 ? 6 : 12;
    QkvAttnArgs a{Xs, widx, b.qkvh ? b.qkvh : b.qkvf, b.qkv_b, b.bias_cmp, wk.zrow, wk.ao, wk.ao_plane, g.nW[res], heads, g.types[res], 0.17677669529663687f};
    if (b.qkvh) return res == 0 ? launch_qa<QaShape<192, 1>>(a, g.types[res], out_planes, s) : launch_qa<QaShape<384, 1>>(a, g.types[res], out_planes, s);
    if (res != 0) return hipErrorNotSupported;             // C = 384 with hi / lo weights: 144 KiB of weights + the bias copies do not fit one CU's LDS
    return launch_qa<QaShape<192, 2>>(a, g.types[res], out_planes, s);
}

template <class P>
hipError_t launch_attention(const AttnArgs<P>& a, hipStream_t stream) {
    constexpr int NPL_O = (P::NA > P::NW ? P::NA : P::NW);
    if (a.bias_cmp) {
        const int types = a.n_win / a.nW;
        if (a.out_planes == 1)
            hipLaunchKernelGGL((earth_attention2_kernel<typename P::T, 1>), dim3((unsigned)(types * a.heads)), dim3(256), 0, stream,
                               a.q, a.k, a.vt, a.bias_cmp, a.out, a.out_plane, a.ld_out, a.nW, a.heads);
        else
            hipLaunchKernelGGL((earth_attention2_kernel<typename P::T, NPL_O>), dim3((unsigned)(types * a.heads)), dim3(256), 0, stream,
                               a.q, a.k, a.vt, a.bias_cmp, a.out, a.out_plane, a.ld_out, a.nW, a.heads);
        return hipGetLastError();
    }
    const long long waves = (long long)a.n_win * a.heads;
    const unsigned blocks = (unsigned)((waves + 3) / 4);
    hipLaunchKernelGGL((earth_attention_kernel<typename P::T, NPL_O>), dim3(blocks), dim3(256), 0, stream,
                       a.q, a.k, a.vt, a.plane, a.bias_exp, a.out, a.out_plane, a.ld_out, a.n_win, a.nW, a.heads);
    return hipGetLastError();
}

template hipError_t launch_attention<PrecBF16x3>(const AttnArgs<PrecBF16x3>&, hipStream_t);
template hipError_t launch_attention<PrecF16x3>(const AttnArgs<PrecF16x3>&, hipStream_t);

}  // namespace skp
