// Attention-side linears of an EarthSpecificBlock in the row-tile form of fused_mlp.hip: the token rows of a wavefront live in
// REGISTERS as MFMA operand fragments for the whole kernel (loaded once, 16 bytes per lane), LDS holds only weights, streamed by
// LDS-DMA in FRAGMENT order (one KiB = what one ds_read_b128 wave instruction consumes, lane-linear, no swizzle) in blocks of 32
// output columns, double-buffered, one barrier per block.  Against the tiled GEMMs of gemm_dma.h this halves the bytes that cross
// LDS (no activation tile) and makes the epilogues wave-local:
//
//   rt_proj_kernel   x[dest(m)] += LayerNorm(norm1)( ao[m] Wp^T + b ),  m over window-ordered rows, dest = window table (reverse +
//                    un-roll + crop; -1 = padding).  3 MFMA terms.  A row's C outputs stay in the wave's accumulators until the end
//                    -> LayerNorm = in-lane sums + two shuffles (no LDS, no barrier); residual rows gathered, 16-byte loads / stores.
//   rt_qkv_kernel    Q (scaled), K -> [win][head][144][32], V -> [win][head][32][144] fp16 from the stream's hi plane (2 MFMA terms:
//                    Q / K / V tolerate an 11-bit A operand).  Rows are gathered through the window table (pad + roll +
//                    partition).  grid.y = 3 selects Q, K or V: one output head (32 columns) per weight block, stored at once.
//                    V runs in the un-swapped operand order (D = X W^T: a lane holds 4 consecutive rows of one column); with the rows
//                    of every 32-row group assigned to fragment rows in perm8 order, the accumulators of a fragment PAIR are 8
//                    consecutive tokens -> 16-byte stores into V^T, as in gemm_dma.h / EpQKV.
//
// gfx950 only.
#include <cstdlib>
#include "gemm_dma.h"
#include "launchers.h"

namespace skp {

__device__ __forceinline__ void rt_ld_pair(const char* p, uint4 (&w)[2]) {
    w[0] = *reinterpret_cast<const uint4*>(p);
    w[1] = *reinterpret_cast<const uint4*>(p + 1024);
    __builtin_amdgcn_sched_barrier(0);          // keep the reads HERE, ahead of the MFMAs that follow (see fused_mlp.hip)
}

// ------------------------------------------------------------------------------------------------------------------------------- //
//  proj + LayerNorm + window reverse + residual
// ------------------------------------------------------------------------------------------------------------------------------- //
template <int C_, int FM_, int NWAVES_, int WPE_>
struct ProjShape {
    static constexpr int C = C_, FM = FM_, NWAVES = NWAVES_, THREADS = 64 * NWAVES_, WPE = WPE_;
    static constexpr int KS = C / 32, CF = C / 16, NCH = C / 32, BM = NWAVES * FM * 16;
    static constexpr int W_BLK = KS * 2 * 2;          // KiB per block of 32 output columns: [ks][n][plane]
    static constexpr int STAGE = W_BLK * 1024;
    static constexpr int SMEM = 2 * STAGE + 3 * C * 4;
    static_assert(W_BLK % NWAVES == 0, "DMA blocks per wave");
};

template <class T>
struct ProjArgs {
    const T* ao; long long ao_plane;       // attention output, window-ordered rows, blocked layout [M/16][C/32][16][32]
    int M;                                 // window rows (multiple of 144)
    T* xs; long long xs_plane;             // residual stream planes
    const int* widx;                       // row -> stream token, -1 = padding
    const T* wf;                           // proj weights in fragment order (prep_rowtile_weights)
    const float *bias, *gamma, *beta;
    float eps;
};

template <class T, class S>
__global__ void __launch_bounds__(S::THREADS) __attribute__((amdgpu_waves_per_eu(S::WPE, S::WPE)))
rt_proj_kernel(const ProjArgs<T> a) {
    constexpr int C = S::C, FM = S::FM, KS = S::KS, CF = S::CF, NCH = S::NCH, NWAVES = S::NWAVES, DEPTH = 3, NS = KS * 2;
    typedef typename OpT<T>::v8 v8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* tab = reinterpret_cast<float*>(smem + 2 * S::STAGE);
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds_base = (unsigned)(size_t)smem;
    auto issue = [&](int j) {
        const T* src = a.wf + ((long long)j * S::W_BLK << 9) + lane * 8;
        const unsigned dst = lds_base + (unsigned)((j & 1) * S::STAGE);
#pragma unroll
        for (int i = 0; i < S::W_BLK / NWAVES; ++i) {
            const int b = wave + i * NWAVES;
            glds16(src + (b << 9), dst + (unsigned)(b << 10));
        }
    };
    issue(0);
    for (int i = tid; i < C; i += S::THREADS) { tab[i] = a.bias[i]; tab[C + i] = a.gamma[i]; tab[2 * C + i] = a.beta[i]; }

    const long long rb0 = (long long)blockIdx.x * (S::BM / 16) + wave * FM;
    v8 xh[FM][KS], xl[FM][KS];
    bool live[FM];
#pragma unroll
    for (int t = 0; t < FM; ++t) {
        live[t] = (rb0 + t) * 16 < a.M;
        const T* p = a.ao + ((live[t] ? rb0 + t : 0) * KS << 9) + l15 * 32 + g * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            xh[t][ks] = *reinterpret_cast<const v8*>(p + (ks << 9));
            xl[t][ks] = *reinterpret_cast<const v8*>(p + (ks << 9) + a.ao_plane);
        }
    }
    // destination rows of the epilogue (window reverse + un-roll + crop): fetched now, their latency hides under the main loop
    int dest[FM];
#pragma unroll
    for (int t = 0; t < FM; ++t) dest[t] = live[t] ? a.widx[(rb0 + t) * 16 + l15] : -1;
#pragma unroll
    for (int t = 0; t < FM; ++t) {
        asm volatile("" : "+v"(dest[t]));
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) { asm volatile("" : "+v"(xh[t][ks])); asm volatile("" : "+v"(xl[t][ks])); }
    }

    f32x4 yacc[FM][CF];
#pragma unroll
    for (int t = 0; t < FM; ++t)
#pragma unroll
        for (int c = 0; c < CF; ++c) yacc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                               // block j landed; every wave is done with block j - 1
        if (j + 1 < NCH) issue(j + 1);
        const char* st = smem + (j & 1) * S::STAGE;
        uint4 ring[DEPTH][2];
#pragma unroll
        for (int s = 0; s < DEPTH - 1 && s < NS; ++s) rt_ld_pair(st + ((s * 2) << 10) + lane * 16, ring[s % DEPTH]);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (s + DEPTH - 1 < NS) rt_ld_pair(st + (((s + DEPTH - 1) * 2) << 10) + lane * 16, ring[(s + DEPTH - 1) % DEPTH]);
            const int ks = s >> 1, c = 2 * j + (s & 1);
            const uint4 wh = ring[s % DEPTH][0], wl = ring[s % DEPTH][1];
#pragma unroll
            for (int t = 0; t < FM; ++t) yacc[t][c] = OpT<T>::mfma(as_v8<T>(wl), xh[t][ks], yacc[t][c]);
#pragma unroll
            for (int t = 0; t < FM; ++t) yacc[t][c] = OpT<T>::mfma(as_v8<T>(wh), xl[t][ks], yacc[t][c]);
#pragma unroll
            for (int t = 0; t < FM; ++t) yacc[t][c] = OpT<T>::mfma(as_v8<T>(wh), xh[t][ks], yacc[t][c]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    const float* tg = tab + C;
    const float* tbe = tab + 2 * C;
#pragma unroll
    for (int t = 0; t < FM; ++t) {
        // residual rows first (gathered 16-byte loads; the input fragments are dead, their registers take the old values)
        const bool ok = dest[t] >= 0;
        const T* old = a.xs + blk_off(ok ? dest[t] : 0, g * 8, C);
        v8 oh[KS], ol[KS];
#pragma unroll
        for (int bp = 0; bp < KS; ++bp) {
            oh[bp] = *reinterpret_cast<const v8*>(old + (bp << 9));
            ol[bp] = *reinterpret_cast<const v8*>(old + (bp << 9) + a.xs_plane);
        }
        float s = 0.f;
#pragma unroll
        for (int bp = 0; bp < KS; ++bp) {
            const int n = 32 * bp + 8 * g;
            const float4 b0 = *reinterpret_cast<const float4*>(tab + n), b1 = *reinterpret_cast<const float4*>(tab + n + 4);
            add8(yacc[t][2 * bp], yacc[t][2 * bp + 1], b0, b1);
#pragma unroll
            for (int r = 0; r < 4; ++r) s += yacc[t][2 * bp][r] + yacc[t][2 * bp + 1][r];
        }
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        const float mean = s * (1.0f / C);
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < CF; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float d = yacc[t][c][r] - mean; q += d * d; }
        q += __shfl_xor(q, 16);
        q += __shfl_xor(q, 32);
        const float rstd = rsqrtf(q * (1.0f / C) + a.eps);
        if (!ok) continue;
        T* dst = a.xs + blk_off(dest[t], g * 8, C);
#pragma unroll
        for (int bp = 0; bp < KS; ++bp) {
            const int n = 32 * bp + 8 * g;
            const float4 g0 = *reinterpret_cast<const float4*>(tg + n), g1 = *reinterpret_cast<const float4*>(tg + n + 4);
            const float4 e0 = *reinterpret_cast<const float4*>(tbe + n), e1 = *reinterpret_cast<const float4*>(tbe + n + 4);
            const f32x4 &x = yacc[t][2 * bp], &z = yacc[t][2 * bp + 1];
            const float y[8] = {(x[0] - mean) * rstd * g0.x + e0.x, (x[1] - mean) * rstd * g0.y + e0.y, (x[2] - mean) * rstd * g0.z + e0.z, (x[3] - mean) * rstd * g0.w + e0.w,
                                (z[0] - mean) * rstd * g1.x + e1.x, (z[1] - mean) * rstd * g1.y + e1.y, (z[2] - mean) * rstd * g1.z + e1.z, (z[3] - mean) * rstd * g1.w + e1.w};
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = ((float)oh[bp][i] + (float)ol[bp][i]) + y[i];
            store8_planes<T, 2>(dst + (bp << 9), a.xs_plane, v);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------- //
//  QKV linear + head split (2 MFMA terms: hi plane of the stream x hi/lo weights)
// ------------------------------------------------------------------------------------------------------------------------------- //
template <int C_, int FM_, int NWAVES_, int NW_ = 2>
struct QkvShape {
    static constexpr int C = C_, FM = FM_, NWAVES = NWAVES_, THREADS = 64 * NWAVES_;
    static constexpr int NW = NW_;             // weight planes: 2 = hi / lo (2 MFMA terms), 1 = hi only (ONE term: the term plan's QKV bit)
    static constexpr int KS = C / 32, HEADS = C / 32, BM = NWAVES * FM * 16;
    static constexpr int W_BLK = KS * 2 * NW, STAGE = W_BLK * 1024;
    static constexpr int SMEM = 2 * STAGE + C * 4;
    static_assert(FM % 2 == 0, "fragment pairs");
};

struct QkvArgs {
    const f16* xs;                         // stream hi plane, blocked layout
    const int* widx;                       // window row -> stream token, -1 = padding
    int M;                                 // window rows (multiple of 144)
    const f16* wf;                         // qkv weights [3C][C] in fragment order, hi/lo
    const float* bias;                     // [3C]
    f16 *q, *k, *vt;
    const f16* zrow;
    float scale;
};

template <class S, bool VPART>
__device__ __forceinline__ void rt_qkv_body(const QkvArgs& a, char* smem) {
    constexpr int C = S::C, FM = S::FM, KS = S::KS, HEADS = S::HEADS, NWAVES = S::NWAVES, DEPTH = 3, NS = KS * 2;
    float* tab = reinterpret_cast<float*>(smem + 2 * S::STAGE);
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int which = VPART ? 2 : (int)blockIdx.y;     // 0 Q, 1 K, 2 V
    const unsigned lds_base = (unsigned)(size_t)smem;
    auto issue = [&](int j) {                          // block j of this part = head j: prepared rows which * C + 32 j ..
        const f16* src = a.wf + ((long long)(which * HEADS + j) * S::W_BLK << 9) + lane * 8;
        const unsigned dst = lds_base + (unsigned)((j & 1) * S::STAGE);
#pragma unroll
        for (int i = 0; i < (S::W_BLK + NWAVES - 1) / NWAVES; ++i) {
            const int b = wave + i * NWAVES;
            if (b < S::W_BLK) glds16(src + (b << 9), dst + (unsigned)(b << 10));
        }
    };
    issue(0);
    for (int i = tid; i < C; i += S::THREADS) tab[i] = a.bias[which * C + i];

    // rows: fragment row 16 (t & 1) + l15 of 32-row group t / 2 holds window row  m0 + 32 (t / 2) + perm8_col(16 (t & 1) + l15)
    const int m0 = blockIdx.x * S::BM + wave * FM * 16;
    const bool tail = m0 + FM * 16 > a.M;              // wave-uniform
    int mrow[FM];
    f16x8 xh[FM][KS];
#pragma unroll
    for (int t = 0; t < FM; ++t) {
        mrow[t] = m0 + 32 * (t >> 1) + perm8_col(16 * (t & 1) + l15);
        const int src = mrow[t] < a.M ? a.widx[mrow[t]] : -1;
        const f16* p = src >= 0 ? a.xs + blk_off(src, g * 8, C) : a.zrow;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) xh[t][ks] = *reinterpret_cast<const f16x8*>(p + (src >= 0 ? (ks << 9) : 0));
    }
#pragma unroll
    for (int t = 0; t < FM; ++t)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(xh[t][ks]));

    for (int j = 0; j < HEADS; ++j) {
        // the DMA of block j is older than the FM stores of block j - 1: waiting down to FM outstanding operations completes it
        // (a wave with rows beyond M skips stores: it waits for everything)
        if (j == 0 || tail) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(FM) : "memory");
        __syncthreads();
        if (j + 1 < HEADS) issue(j + 1);
        const char* st = smem + (j & 1) * S::STAGE;
        f32x4 acc[FM][2];
#pragma unroll
        for (int t = 0; t < FM; ++t) { acc[t][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[t][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        uint4 ring[DEPTH][2];
        if constexpr (S::NW == 1) {
            // one weight plane: a ring step is the fragment PAIR (ks, n = 0 / 1) = two consecutive KiB, one MFMA term each
#pragma unroll
            for (int s = 0; s < DEPTH - 1 && s < KS; ++s) rt_ld_pair(st + ((s * 2) << 10) + lane * 16, ring[s % DEPTH]);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if (ks + DEPTH - 1 < KS) rt_ld_pair(st + (((ks + DEPTH - 1) * 2) << 10) + lane * 16, ring[(ks + DEPTH - 1) % DEPTH]);
                const f16x8 w0 = as_v8<f16>(ring[ks % DEPTH][0]), w1 = as_v8<f16>(ring[ks % DEPTH][1]);
                if constexpr (!VPART) {
#pragma unroll
                    for (int t = 0; t < FM; ++t) acc[t][0] = OpT<f16>::mfma(w0, xh[t][ks], acc[t][0]);
#pragma unroll
                    for (int t = 0; t < FM; ++t) acc[t][1] = OpT<f16>::mfma(w1, xh[t][ks], acc[t][1]);
                } else {
#pragma unroll
                    for (int t = 0; t < FM; ++t) acc[t][0] = OpT<f16>::mfma(xh[t][ks], w0, acc[t][0]);
#pragma unroll
                    for (int t = 0; t < FM; ++t) acc[t][1] = OpT<f16>::mfma(xh[t][ks], w1, acc[t][1]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
        for (int s = 0; s < DEPTH - 1 && s < NS; ++s) rt_ld_pair(st + ((s * 2) << 10) + lane * 16, ring[s % DEPTH]);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (s + DEPTH - 1 < NS) rt_ld_pair(st + (((s + DEPTH - 1) * 2) << 10) + lane * 16, ring[(s + DEPTH - 1) % DEPTH]);
            const int ks = s >> 1, n = s & 1;
            const f16x8 wh = as_v8<f16>(ring[s % DEPTH][0]), wl = as_v8<f16>(ring[s % DEPTH][1]);
            if constexpr (!VPART) {                    // swapped: D^T = W X^T, a lane holds 4 consecutive columns of one row
#pragma unroll
                for (int t = 0; t < FM; ++t) acc[t][n] = OpT<f16>::mfma(wl, xh[t][ks], acc[t][n]);
#pragma unroll
                for (int t = 0; t < FM; ++t) acc[t][n] = OpT<f16>::mfma(wh, xh[t][ks], acc[t][n]);
            } else {                                   // un-swapped: D = X W^T, a lane holds 4 consecutive rows of one column
#pragma unroll
                for (int t = 0; t < FM; ++t) acc[t][n] = OpT<f16>::mfma(xh[t][ks], wl, acc[t][n]);
#pragma unroll
                for (int t = 0; t < FM; ++t) acc[t][n] = OpT<f16>::mfma(xh[t][ks], wh, acc[t][n]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        }
        if constexpr (!VPART) {
            // prepared rows 16 n + 4 g + r of the block are output columns 8 g + 4 n + r (perm8): 8 consecutive d of head j
            const float4 b0 = *reinterpret_cast<const float4*>(tab + 32 * j + 8 * g), b1 = *reinterpret_cast<const float4*>(tab + 32 * j + 8 * g + 4);
            const float sc = which == 0 ? a.scale : 1.0f;
            f16* base = which == 0 ? a.q : a.k;
#pragma unroll
            for (int t = 0; t < FM; ++t) {
                const int m = mrow[t];
                const int win = m / WIN_TOKENS, tt = m - win * WIN_TOKENS;
                const f32x4 &x = acc[t][0], &y = acc[t][1];
                const float v[8] = {(x[0] + b0.x) * sc, (x[1] + b0.y) * sc, (x[2] + b0.z) * sc, (x[3] + b0.w) * sc,
                                    (y[0] + b1.x) * sc, (y[1] + b1.y) * sc, (y[2] + b1.z) * sc, (y[3] + b1.w) * sc};
                f16* dst = base + (((long long)win * HEADS + j) * WIN_TOKENS + tt) * HEAD_DIM + 8 * g;
                if (m < a.M) store8_planes<f16, 1>(dst, 0, v);
            }
        } else {
            // a lane's column: prepared row 16 n + l15 -> d = perm8_col(16 n + l15); its rows 4 g + r of fragment t are the window rows
            // m0 + 32 (t / 2) + 8 g + 4 (t & 1) + r: a fragment pair = 8 consecutive tokens
#pragma unroll
            for (int tp = 0; tp < FM / 2; ++tp) {
                const int m = m0 + 32 * tp + 8 * g;
                const int win = m / WIN_TOKENS, tt = m - win * WIN_TOKENS;     // 144 % 8 == 0: the 8 tokens share a window
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const int d = perm8_col(16 * n + l15);
                    const float bb = tab[32 * j + d];
                    const f32x4 &x = acc[2 * tp][n], &y = acc[2 * tp + 1][n];
                    const float v[8] = {x[0] + bb, x[1] + bb, x[2] + bb, x[3] + bb, y[0] + bb, y[1] + bb, y[2] + bb, y[3] + bb};
                    f16* dst = a.vt + (((long long)win * HEADS + j) * HEAD_DIM + d) * WIN_TOKENS + tt;
                    if (m < a.M) store8_planes<f16, 1>(dst, 0, v);
                }
            }
        }
    }
}

template <class S>
__global__ void __launch_bounds__(S::THREADS) rt_qkv_kernel(const QkvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (blockIdx.y == 2) rt_qkv_body<S, true>(a, smem);
    else rt_qkv_body<S, false>(a, smem);
}

// ---- prepare: [N][K] fp32 -> fragment-order hi/lo planes, rows in perm8 order -------------------------------------------------- //
//   wf[((j KS + ks) 2 + n) 2 + plane][lane][e] = W[perm8_col(32 j + 16 n + (lane & 15))][32 ks + 8 (lane >> 4) + e]
template <class T>
__global__ void prep_rowtile_kernel(const float* __restrict__ w, T* __restrict__ out, int N, int K, int planes) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int KS = K / 32;
    const long long total = (long long)(N / 32) * KS * 2 * 512;
    if (i >= total) return;
    const int e = (int)(i & 7), lane = (int)((i >> 3) & 63);
    long long q = i >> 9;
    const int n = (int)(q & 1); q >>= 1;
    const int ks = (int)(q % KS);
    const int j = (int)(q / KS);
    const float v = w[(long long)perm8_col(32 * j + 16 * n + (lane & 15)) * K + 32 * ks + 8 * (lane >> 4) + e];
    const T h = (T)v;
    const long long o = ((((long long)j * KS + ks) * 2 + n) * planes << 9) + lane * 8 + e;
    out[o] = h;
    if (planes == 2) out[o + 512] = (T)(v - (float)h);
}

template <class T>
hipError_t prep_rowtile_weights(const float* w, T* wf, int N, int K, hipStream_t s, int planes) {
    if ((N & 31) || (K & 31) || (planes != 1 && planes != 2)) return hipErrorInvalidValue;
    const long long total = (long long)(N / 32) * (K / 32) * 2 * 512;
    hipLaunchKernelGGL((prep_rowtile_kernel<T>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, wf, N, K, planes);
    return hipGetLastError();
}
template hipError_t prep_rowtile_weights<bf16>(const float*, bf16*, int, int, hipStream_t, int);
template hipError_t prep_rowtile_weights<f16>(const float*, f16*, int, int, hipStream_t, int);

template <class T, class S>
static hipError_t launch_proj(const ProjArgs<T>& a, hipStream_t s) {
    auto kern = rt_proj_kernel<T, S>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, S::SMEM);
    if (e != hipSuccess) return e;
    const unsigned grid = (unsigned)((a.M + S::BM - 1) / S::BM);
    if (grid == 0) return hipSuccess;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(S::THREADS), S::SMEM, s, a);
    return hipGetLastError();
}

template <class P>
hipError_t op_proj_rowtile(const Geom& g, const BlockW<typename P::T>& b, const int* widx, int res, typename P::T* Xs, const Work<P>& wk, hipStream_t s) {
    typedef typename P::T T;
    static_assert(P::NA == 2 && P::NW == 2, "3-term path");
    ProjArgs<T> a{wk.ao, wk.ao_plane, g.mwin[res], Xs, wk.xs_plane[res], widx, b.projf, b.proj_b, b.n1_g, b.n1_b, 1e-5f};
    // C = 192: two 4-wave workgroups per CU, 32 rows per wave;  C = 384: one 8-wave workgroup, 16 rows per wave (fused_mlp.hip's findings)
    if (res == 0) return launch_proj<T, ProjShape<192, 2, 4, 2>>(a, s);
    return launch_proj<T, ProjShape<384, 1, 8, 2>>(a, s);
}
template hipError_t op_proj_rowtile<PrecBF16x3>(const Geom&, const BlockW<bf16>&, const int*, int, bf16*, const Work<PrecBF16x3>&, hipStream_t);
template hipError_t op_proj_rowtile<PrecF16x3>(const Geom&, const BlockW<f16>&, const int*, int, f16*, const Work<PrecF16x3>&, hipStream_t);

template <class S>
static hipError_t launch_qkv(const QkvArgs& a, hipStream_t s) {
    auto kern = rt_qkv_kernel<S>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, S::SMEM);
    if (e != hipSuccess) return e;
    const unsigned grid = (unsigned)((a.M + S::BM - 1) / S::BM);
    if (grid == 0) return hipSuccess;
    hipLaunchKernelGGL(kern, dim3(grid, 3), dim3(S::THREADS), S::SMEM, s, a);
    return hipGetLastError();
}

// fp16 planes, QKV from the hi plane only: 2 terms against hi / lo weights (b.qkvf), or ONE term against the hi plane (b.qkvh)
hipError_t op_qkv_rowtile(const Geom& g, const BlockW<f16>& b, const int* widx, int res, const f16* Xs, const Work<PrecF16x3>& wk, hipStream_t s) {
    QkvArgs a{Xs, widx, g.mwin[res], b.qkvh ? b.qkvh : b.qkvf, b.qkv_b, wk.q, wk.k, wk.vt, wk.zrow, 0.17677669529663687f};
    if (b.qkvh) {
        if (res == 0) return launch_qkv<QkvShape<192, 2, 8, 1>>(a, s);
        return launch_qkv<QkvShape<384, 2, 8, 1>>(a, s);
    }
    if (res == 0) return launch_qkv<QkvShape<192, 2, 8>>(a, s);
    return launch_qkv<QkvShape<384, 2, 8>>(a, s);
}

}  // namespace skp
