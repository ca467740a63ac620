// GraphCast interaction-network updates as ONE kernel each (include/skyrim_graphcast.h: skgc_edge_update, skgc_node_mlp).
//
// An edge update of the reference's typed graph network (/root/reference/skyrim/core/models/graphcast.py:118 -> DeepMind's
// InteractionNetwork) is   e' = e + LayerNorm(W2 swish(W1 concat(e, v_s[send], v_r[recv]) + b1) + b2),   agg[recv] += (e' - e).
// By distributivity W1 concat(...) = e W_e^T + (v_s W_s^T)[send] + (v_r W_r^T)[recv]: the node terms are computed once per NODE by a plain
// GEMM; here the rest runs for a tile of 128 edge rows without leaving the CU:
//
//   * a wavefront owns 2 x 16 rows.  Their 512 input columns sit in REGISTERS as MFMA B-operand fragments (one fp16 plane, 16-byte
//     loads of the blocked layout: one wave instruction = one contiguous 1 KiB block), the weights stream through LDS in FRAGMENT
//     order by LDS-DMA (64 KiB stages, two stages, one barrier per stage), as fp16 hi/lo planes: two MFMA terms W_hi x + W_lo x.
//     Rounding the ACTIVATIONS of the edge MLPs to one fp16 plane costs 1.7e-4 of the predicted increment at production width and
//     depth (tools/graphcast_numerics.py); rounding their WEIGHTS would cost 4.3e-4, so the weights keep both planes.
//   * phase 1: the first Linear in chunks of 32 hidden units.  The accumulators START from the gathered node terms (stored in "pos"
//     column order, skyrim_amd/graphcast/fused.py: a lane's 8 values are 32 contiguous bytes; the gathers of chunk j + 1 fly during
//     chunk j) and END, after swish and the fp16 rounding, as the second Linear's B-operand fragments -- in swapped order
//     (D^T = W X^T) a lane's accumulators ARE its k-slots of the next GEMM (same trick as fused_mlp.hip).  The swish of chunk j - 1 is
//     spliced between the MFMAs of chunk j.
//   * phase 2: the second Linear, 16 k-steps x 32 output fragments into 2 x 128 accumulator registers; perm8 weight rows make a
//     fragment pair 8 consecutive output columns.
//   * epilogue: LayerNorm from the accumulators (in-lane sums + two shuffles), the residual update written back in the blocked
//     layout, and the RECEIVER SUM from the same registers: rows are sorted by receiver, so a segmented scan over the 16 lanes of a
//     DPP row (4 steps) leaves every run's sum in its last row; runs that cross a wave's 16-row groups are completed through a 16 KiB
//     LDS exchange in a fixed order (deterministic, no atomics).  The host packs the rows so that a run never crosses a 128-row tile
//     unless it is longer than a tile; the pieces of such a run go to a side buffer and skgc_segment_fixup adds them in order.
//
// The `static` form (grid->mesh, mesh->grid: edge latents that do not depend on the input) has no first Linear at all: its phase 1 is
// swish(prepared term + gathered node terms).  The node form (skgc_node_mlp) keeps fp32 node latents exact: hi/lo fragments made on
// the fly from fp32 rows, three MFMA terms, one 16-row group per wave.
//
// Per tile of 128 rows: 2 MiB (1 MiB static) of weights through LDS-DMA against 65.5 k (32.8 k) MFMA clocks per SIMD = 32 B/clk/CU
// (measured ceiling from L2: 43, tools/micro/dma_bw.hip); every weight fragment read from LDS feeds two MFMAs (128 B/clk/CU of 256).
// gfx950 only.  One workgroup (4 waves, one per SIMD, 512 registers) per CU; grid = tiles.
#include <cstdlib>
#include "../../include/skyrim_graphcast.h"
#include "gemm_dma.h"

namespace skp {

__device__ __forceinline__ float swish_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }

constexpr int FZ_L = 512, FZ_KS = 16, FZ_CF = 32, FZ_NCH = 16, FZ_TILE = 128;
constexpr int FZ_STAGE = 64 * 1024;
constexpr int FZ_SMEM = 2 * FZ_STAGE + 4 * FZ_L * 4;

struct EdgeArgs {
    const f16* e_in;        // FC1: edge latents; static: prepared first-Linear term ("pos" columns).  Blocked fp16 [rows][512]
    f16* e_out;             // nullable (FC1 only): e_in + LayerNorm(...), blocked fp16; may alias e_in
    const float* term[2];   // gathered node terms, fp32 rows in "pos" column order
    const int* idx[2];      // node of every packed row (< 0: padding row)
    long long ld[2];
    const int* recv;        // receiver of every packed row (< 0: padding row); rows sorted by receiver
    const f16* w1f;         // FC1: first Linear (edge part), fragment order, hi/lo planes
    const f16* w2f;         // second Linear, fragment order, hi/lo planes
    const float *b2, *gamma, *beta;
    float* agg;             // [nodes][512] receiver sums
    float* heads;           // [tiles][512] continuation pieces (runs longer than a tile)
    float eps;
};

// DEPTH register pairs (hi, lo plane of one fragment) read DEPTH - 1 pairs ahead of their MFMAs; the scheduling barrier pins the reads
// in program order (fused_mlp.hip: ld_pair)
__device__ __forceinline__ void fz_ld2(const char* p, uint4 (&w)[2]) {
    w[0] = *reinterpret_cast<const uint4*>(p);
    w[1] = *reinterpret_cast<const uint4*>(p + 1024);
    __builtin_amdgcn_sched_barrier(0);
}
template <int NP, int RD, class Body>
__device__ __forceinline__ void fz_stream(const char* st, Body&& body) {
    uint4 ring[RD][2];
#pragma unroll
    for (int p = 0; p < RD - 1 && p < NP; ++p) fz_ld2(st + (p << 11), ring[p % RD]);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        if (p + RD - 1 < NP) fz_ld2(st + ((p + RD - 1) << 11), ring[(p + RD - 1) % RD]);
        body(p, ring[p % RD][0], ring[p % RD][1]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// 64 KiB of fragments -> one LDS stage: block q by wave q % 4
__device__ __forceinline__ void fz_dma64k(const f16* src, unsigned dst, int wave, int lane) {
    const f16* s = src + lane * 8;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int q = wave + i * 4;
        glds16(s + (q << 9), dst + (unsigned)(q << 10));
    }
}

__device__ __forceinline__ f16x8 fz_pack8(const f32x4& a, const f32x4& b) {
    f16x8 h;
#pragma unroll
    for (int i = 0; i < 4; ++i) { h[i] = (f16)a[i]; h[4 + i] = (f16)b[i]; }
    return h;
}

template <int CTRL>
__device__ __forceinline__ float fz_dpp(float v) {       // row_shr within the 16 lanes of a DPP row; lanes without a source read 0
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ int fz_dpp_i(int v, int old) { // ... lanes without a source keep `old`
    return __builtin_amdgcn_update_dpp(old, v, CTRL, 0xf, 0xf, false);
}

// ---- LayerNorm of a 16-row group held as 32 accumulator fragments (perm8: pair bp = columns 32 bp + 8 g + [0..7]) ------------------ //
__device__ __forceinline__ void fz_layer_norm(f32x4 (&y)[FZ_CF], const float* tab, int g, float eps) {
    const float *tb = tab, *tg = tab + FZ_L, *te = tab + 2 * FZ_L;
    float s = 0.f;
#pragma unroll
    for (int bp = 0; bp < FZ_KS; ++bp) {
        const int n = 32 * bp + 8 * g;
        const float4 b0 = *reinterpret_cast<const float4*>(tb + n), b1 = *reinterpret_cast<const float4*>(tb + n + 4);
        add8(y[2 * bp], y[2 * bp + 1], b0, b1);
#pragma unroll
        for (int r = 0; r < 4; ++r) s += y[2 * bp][r] + y[2 * bp + 1][r];
    }
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    const float mean = s * (1.0f / FZ_L);
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < FZ_CF; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float d = y[c][r] - mean; q += d * d; }
    q += __shfl_xor(q, 16);
    q += __shfl_xor(q, 32);
    const float rstd = rsqrtf(q * (1.0f / FZ_L) + eps);
#pragma unroll
    for (int bp = 0; bp < FZ_KS; ++bp) {
        const int n = 32 * bp + 8 * g;
        const float4 g0 = *reinterpret_cast<const float4*>(tg + n), g1 = *reinterpret_cast<const float4*>(tg + n + 4);
        const float4 e0 = *reinterpret_cast<const float4*>(te + n), e1 = *reinterpret_cast<const float4*>(te + n + 4);
        f32x4 &x = y[2 * bp], &z = y[2 * bp + 1];
        x[0] = (x[0] - mean) * rstd * g0.x + e0.x; x[1] = (x[1] - mean) * rstd * g0.y + e0.y;
        x[2] = (x[2] - mean) * rstd * g0.z + e0.z; x[3] = (x[3] - mean) * rstd * g0.w + e0.w;
        z[0] = (z[0] - mean) * rstd * g1.x + e1.x; z[1] = (z[1] - mean) * rstd * g1.y + e1.y;
        z[2] = (z[2] - mean) * rstd * g1.z + e1.z; z[3] = (z[3] - mean) * rstd * g1.w + e1.w;
    }
}

template <bool FC1, int NT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
edge_update_kernel(const EdgeArgs a) {
    constexpr int FM = 2, KS = FZ_KS, CF = FZ_CF, NCH = FZ_NCH, RD = 3;
    typedef typename OpT<f16>::v8 v8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* tab = reinterpret_cast<float*>(smem + 2 * FZ_STAGE);
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds_base = (unsigned)(size_t)smem;
    const char* lrd = smem + lane * 16;
    const long long tile0 = (long long)blockIdx.x * FZ_TILE;

    // ---- prologue: first weight stages, tables, rows ------------------------------------------------------------------------------ //
    if constexpr (FC1) {
        fz_dma64k(a.w1f, lds_base, wave, lane);
    } else {
        fz_dma64k(a.w2f, lds_base, wave, lane);
        fz_dma64k(a.w2f + (FZ_STAGE / 2), lds_base + FZ_STAGE, wave, lane);
    }
    for (int i = tid; i < FZ_L; i += 256) { tab[i] = a.b2[i]; tab[FZ_L + i] = a.gamma[i]; tab[2 * FZ_L + i] = a.beta[i]; }

    long long row[FM];
    int my[FM];
    const float* tp[FM][NT > 0 ? NT : 1];
#pragma unroll
    for (int t = 0; t < FM; ++t) {
        row[t] = tile0 + wave * 32 + t * 16 + l15;
        my[t] = a.recv[row[t]];
#pragma unroll
        for (int s = 0; s < NT; ++s) {
            const int n = a.idx[s][row[t]];
            tp[t][s] = a.term[s] + (long long)(n < 0 ? 0 : n) * a.ld[s] + 8 * g;
        }
    }
    const long long rb0 = (tile0 >> 4) + wave * 2;                    // first 16-row block of the wave

    f16x8 hh[FM][NCH];                                                // the second Linear's operand: hidden activations, one fp16 plane
    f32x4 pre[FM][2];                                                 // pre-activations of the chunk whose swish is in flight

    auto swish4 = [&](f32x4& v) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = swish_f(v[r]);
    };

    if constexpr (FC1) {
        v8 xh[FM][KS];
#pragma unroll
        for (int t = 0; t < FM; ++t) {
            const f16* p = a.e_in + (((rb0 + t) * KS) << 9) + l15 * 32 + g * 8;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) xh[t][ks] = *reinterpret_cast<const v8*>(p + (ks << 9));
        }
        f32x4 gat[FM][NT > 0 ? NT : 1][2];
        auto gather = [&](int j) {
#pragma unroll
            for (int t = 0; t < FM; ++t)
#pragma unroll
                for (int s = 0; s < NT; ++s) {
                    gat[t][s][0] = *reinterpret_cast<const f32x4*>(tp[t][s] + 32 * j);
                    gat[t][s][1] = *reinterpret_cast<const f32x4*>(tp[t][s] + 32 * j + 4);
                }
        };
        gather(0);
        // consumed once here so that hipcc's vmcnt waits for these loads sit BEFORE the loop (inside it they would also wait for the
        // LDS-DMA in flight, which the compiler's counter bookkeeping does not know about; fused_mlp.hip)
#pragma unroll
        for (int t = 0; t < FM; ++t)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(xh[t][ks]));

        f32x4 hacc[FM][2];
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // the gathered terms of chunk j have landed with everything else: hand them to the compiler HERE, before the next requests
#pragma unroll
            for (int t = 0; t < FM; ++t)
#pragma unroll
                for (int s = 0; s < NT; ++s) { asm volatile("" : "+v"(gat[t][s][0])); asm volatile("" : "+v"(gat[t][s][1])); }
            __syncthreads();                           // W1(j) landed in stage j & 1; every wave is done with the other stage
            if (j > 0) {
#pragma unroll
                for (int t = 0; t < FM; ++t) { pre[t][0] = hacc[t][0]; pre[t][1] = hacc[t][1]; }
            }
#pragma unroll
            for (int t = 0; t < FM; ++t) {
                f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < NT; ++s) {
                    lo += gat[t][s][0];
                    hi += gat[t][s][1];
                }
                hacc[t][0] = lo; hacc[t][1] = hi;
            }
            asm volatile("" ::: "memory");
            if (j + 1 < NCH) fz_dma64k(a.w1f + (long long)(j + 1) * (FZ_STAGE / 2), lds_base + ((j + 1) & 1) * FZ_STAGE, wave, lane);
            else fz_dma64k(a.w2f, lds_base, wave, lane);                 // chunk 15 reads stage 1: stage 0 is free for W2(0)
            if (j + 1 < NCH) gather(j + 1);
            // pair p = (n, ks): W1 rows 32 j + 16 n + [0, 16), k-step ks; the swish of chunk j - 1 rides between the MFMAs
            fz_stream<2 * KS, RD>(lrd + (j & 1) * FZ_STAGE, [&](int p, const uint4& wh, const uint4& wl) {
                const int n = p >> 4, ks = p & 15;
#pragma unroll
                for (int t = 0; t < FM; ++t) hacc[t][n] = OpT<f16>::mfma(as_v8<f16>(wl), xh[t][ks], hacc[t][n]);
#pragma unroll
                for (int t = 0; t < FM; ++t) hacc[t][n] = OpT<f16>::mfma(as_v8<f16>(wh), xh[t][ks], hacc[t][n]);
                if (j > 0) {
                    if (p == 4) swish4(pre[0][0]);
                    if (p == 8) swish4(pre[0][1]);
                    if (p == 12) swish4(pre[1][0]);
                    if (p == 16) swish4(pre[1][1]);
                    if (p == 20) hh[0][j > 0 ? j - 1 : 0] = fz_pack8(pre[0][0], pre[0][1]);
                    if (p == 24) hh[1][j > 0 ? j - 1 : 0] = fz_pack8(pre[1][0], pre[1][1]);
                }
            });
        }
#pragma unroll
        for (int t = 0; t < FM; ++t) { pre[t][0] = hacc[t][0]; pre[t][1] = hacc[t][1]; }
    } else {
        // static form: pre-activation = prepared term (fp16, "pos" columns: the lane's 16 bytes of block (row block, j)) + gathered node terms
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
#pragma unroll
            for (int t = 0; t < FM; ++t) {
                const v8 st = *reinterpret_cast<const v8*>(a.e_in + ((((rb0 + t) * KS) + j) << 9) + l15 * 32 + g * 8);
                f32x4 lo = {(float)st[0], (float)st[1], (float)st[2], (float)st[3]}, hi = {(float)st[4], (float)st[5], (float)st[6], (float)st[7]};
#pragma unroll
                for (int s = 0; s < NT; ++s) {
                    const float4 u0 = *reinterpret_cast<const float4*>(tp[t][s] + 32 * j), u1 = *reinterpret_cast<const float4*>(tp[t][s] + 32 * j + 4);
                    lo[0] += u0.x; lo[1] += u0.y; lo[2] += u0.z; lo[3] += u0.w;
                    hi[0] += u1.x; hi[1] += u1.y; hi[2] += u1.z; hi[3] += u1.w;
                }
                swish4(lo); swish4(hi);
                hh[t][j] = fz_pack8(lo, hi);
            }
        }
#pragma unroll
        for (int t = 0; t < FM; ++t)
#pragma unroll
            for (int j = 0; j < NCH; ++j) asm volatile("" : "+v"(hh[t][j]));
    }

    // ---- phase 2: second Linear ----------------------------------------------------------------------------------------------------- //
    f32x4 yacc[FM][CF];
#pragma unroll
    for (int t = 0; t < FM; ++t)
#pragma unroll
        for (int c = 0; c < CF; ++c) yacc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    if constexpr (FC1) {                                // the last chunk's swish has nothing left to hide under: do it here
#pragma unroll
        for (int t = 0; t < FM; ++t) { swish4(pre[t][0]); swish4(pre[t][1]); hh[t][NCH - 1] = fz_pack8(pre[t][0], pre[t][1]); }
    }
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                               // W2(j) landed in stage j & 1; every wave is done with the other stage
        if (j + 1 < NCH && (FC1 || j > 0)) fz_dma64k(a.w2f + (long long)(j + 1) * (FZ_STAGE / 2), lds_base + ((j + 1) & 1) * FZ_STAGE, wave, lane);
        fz_stream<CF, RD>(lrd + (j & 1) * FZ_STAGE, [&](int c, const uint4& wh, const uint4& wl) {
#pragma unroll
            for (int t = 0; t < FM; ++t) yacc[t][c] = OpT<f16>::mfma(as_v8<f16>(wl), hh[t][j], yacc[t][c]);
#pragma unroll
            for (int t = 0; t < FM; ++t) yacc[t][c] = OpT<f16>::mfma(as_v8<f16>(wh), hh[t][j], yacc[t][c]);
        });
    }

    // ---- epilogue: LayerNorm, residual update, receiver sum ---------------------------------------------------------------------- //
    const int prv0 = tile0 > 0 ? a.recv[tile0 - 1] : -2;              // wave-uniform: does the tile's first run continue the previous tile's?
    const int first = a.recv[tile0];
    const bool tile_cont = first >= 0 && prv0 == first;
    int nxt[FM];
#pragma unroll
    for (int t = 0; t < FM; ++t) nxt[t] = (wave * 32 + t * 16 + l15 + 1 < FZ_TILE) ? a.recv[row[t] + 1] : -3;
    // per 16-row group: LayerNorm, residual update, then the segmented inclusive scan over the group's rows -- after it the LAST row of
    // every run holds the run's sum inside the group
#pragma unroll
    for (int t = 0; t < FM; ++t) {
        bool stored = false;
        if constexpr (FC1) {
            if (a.e_out != nullptr) {
                const long long off = (((rb0 + t) * KS) << 9) + l15 * 32 + g * 8;
                v8 xr[KS];
#pragma unroll
                for (int bp = 0; bp < KS; ++bp) xr[bp] = *reinterpret_cast<const v8*>(a.e_in + off + (bp << 9));
                fz_layer_norm(yacc[t], tab, g, a.eps);
                stored = true;
                if (my[t] >= 0) {
#pragma unroll
                    for (int bp = 0; bp < KS; ++bp) {
                        f32x4 lo, hi;
#pragma unroll
                        for (int i = 0; i < 4; ++i) { lo[i] = (float)xr[bp][i] + yacc[t][2 * bp][i]; hi[i] = (float)xr[bp][4 + i] + yacc[t][2 * bp + 1][i]; }
                        *reinterpret_cast<f16x8*>(a.e_out + off + (bp << 9)) = fz_pack8(lo, hi);
                    }
                }
            }
        }
        if (!stored) fz_layer_norm(yacc[t], tab, g, a.eps);
        const int m = my[t];
        const float m1 = (m >= 0 && fz_dpp_i<0x111>(m, -7) == m) ? 1.f : 0.f, m2 = (m >= 0 && fz_dpp_i<0x112>(m, -7) == m) ? 1.f : 0.f;
        const float m4 = (m >= 0 && fz_dpp_i<0x114>(m, -7) == m) ? 1.f : 0.f, m8 = (m >= 0 && fz_dpp_i<0x118>(m, -7) == m) ? 1.f : 0.f;
#pragma unroll
        for (int c = 0; c < CF; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = yacc[t][c][r];
                v = fmaf(fz_dpp<0x111>(v), m1, v);
                v = fmaf(fz_dpp<0x112>(v), m2, v);
                v = fmaf(fz_dpp<0x114>(v), m4, v);
                v = fmaf(fz_dpp<0x118>(v), m8, v);
                yacc[t][c][r] = v;
            }
    }
    // exchange between the 8 groups of the tile: group u = 2 wave + t publishes the sum of its LAST run (row 15) and that run's receiver
    __syncthreads();                                   // every wave is done with the weight stages: stage 0 becomes the exchange buffer
    float* tails = reinterpret_cast<float*>(smem);     // [8 groups][4 g][128]
    int* tmeta = reinterpret_cast<int*>(smem + 8 * 2048);
#pragma unroll
    for (int t = 0; t < FM; ++t) {
        const int u = wave * 2 + t;
        if (l15 == 15) {
            float* dst = tails + u * 512 + g * 128;
#pragma unroll
            for (int bp = 0; bp < KS; ++bp) {
                *reinterpret_cast<float4*>(dst + bp * 8) = make_float4(yacc[t][2 * bp][0], yacc[t][2 * bp][1], yacc[t][2 * bp][2], yacc[t][2 * bp][3]);
                *reinterpret_cast<float4*>(dst + bp * 8 + 4) = make_float4(yacc[t][2 * bp + 1][0], yacc[t][2 * bp + 1][1], yacc[t][2 * bp + 1][2], yacc[t][2 * bp + 1][3]);
            }
            if (g == 0) tmeta[u] = my[t];
        }
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < FM; ++t) {
        const int u = wave * 2 + t;
        const int head = __builtin_amdgcn_readlane(my[t], 0);         // receiver of the group's first row (wave-uniform)
        const bool in_head = my[t] == head;
        // the groups before this one whose last run is this group's first run, nearest first (a fixed order: the sum is deterministic)
        for (int up = u - 1; up >= 0 && head >= 0 && tmeta[up] == head; --up) {
            const float* src = tails + up * 512 + g * 128;
#pragma unroll
            for (int bp = 0; bp < KS; ++bp) {
                const float4 c0 = *reinterpret_cast<const float4*>(src + bp * 8), c1 = *reinterpret_cast<const float4*>(src + bp * 8 + 4);
                if (in_head) {
                    yacc[t][2 * bp][0] += c0.x; yacc[t][2 * bp][1] += c0.y; yacc[t][2 * bp][2] += c0.z; yacc[t][2 * bp][3] += c0.w;
                    yacc[t][2 * bp + 1][0] += c1.x; yacc[t][2 * bp + 1][1] += c1.y; yacc[t][2 * bp + 1][2] += c1.z; yacc[t][2 * bp + 1][3] += c1.w;
                }
            }
        }
        // a run ends where the next row has another receiver (or the tile ends): its last row writes the sum
        if (my[t] >= 0 && my[t] != nxt[t]) {
            float* dst = (tile_cont && my[t] == first) ? a.heads + (long long)blockIdx.x * FZ_L : a.agg + (long long)my[t] * FZ_L;
            dst += 8 * g;
#pragma unroll
            for (int bp = 0; bp < KS; ++bp) {
                *reinterpret_cast<float4*>(dst + 32 * bp) = make_float4(yacc[t][2 * bp][0], yacc[t][2 * bp][1], yacc[t][2 * bp][2], yacc[t][2 * bp][3]);
                *reinterpret_cast<float4*>(dst + 32 * bp + 4) = make_float4(yacc[t][2 * bp + 1][0], yacc[t][2 * bp + 1][1], yacc[t][2 * bp + 1][2], yacc[t][2 * bp + 1][3]);
            }
        }
    }
}

// agg[node[i]] += heads[tiles[first[i]]] + heads[tiles[first[i] + 1]] + ...   in that order; one workgroup of 128 lanes x float4 per node
__global__ void __launch_bounds__(128) segment_fixup_kernel(float* __restrict__ agg, const float* __restrict__ heads, const int* __restrict__ nodes,
                                                            const int* __restrict__ first, const int* __restrict__ tiles) {
    const int i = blockIdx.x, c = threadIdx.x * 4;
    float4 s = *reinterpret_cast<const float4*>(agg + (long long)nodes[i] * FZ_L + c);
    for (int k = first[i]; k < first[i + 1]; ++k) {
        const float4 h = *reinterpret_cast<const float4*>(heads + (long long)tiles[k] * FZ_L + c);
        s.x += h.x; s.y += h.y; s.z += h.z; s.w += h.w;
    }
    *reinterpret_cast<float4*>(agg + (long long)nodes[i] * FZ_L + c) = s;
}

// ---- node update:  out = res + LayerNorm(W2 swish(W1 concat(src...) + b1) + b2)  on fp32 rows, three MFMA terms ------------------------ //
// The node latents are the network's residual streams; rounding them (or the node MLPs' hidden activations) to one fp16 plane costs
// 4-5e-4 of the predicted increment (tools/graphcast_numerics.py), so this form splits every fp32 operand into fp16 hi/lo fragments on the
// fly: A W^T ~ A_hi W_lo^T + A_lo W_hi^T + A_hi W_hi^T.  One 16-row group per wave (x: 128 / 256 registers, hidden: 128), 64 rows per tile.
struct NodeArgs {
    const float* src[2];    // fp32 rows, 512 columns each (concatenated along K)
    long long ld[2];
    const f16 *w1f, *w2f;   // fragment order, hi/lo planes: [512][512 n_src], [512][512]
    const float *b1, *b2, *gamma, *beta;
    const float* res;       // nullable; out may alias res
    long long ld_res;
    float* out;
    long long ld_out;
    long long rows;
    float eps;
};

template <int NS>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
node_mlp_kernel(const NodeArgs a) {
    constexpr int KS = FZ_KS * NS, CF = FZ_CF, NCH = FZ_NCH, RD = 3;
    constexpr int NST = NCH * NS;                                      // phase-1 stages of 64 KiB: a chunk of 32 units (NS = 1) or half a chunk
    typedef typename OpT<f16>::v8 v8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* tab = reinterpret_cast<float*>(smem + 2 * FZ_STAGE);        // b2 | gamma | beta | b1
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds_base = (unsigned)(size_t)smem;
    const char* lrd = smem + lane * 16;

    fz_dma64k(a.w1f, lds_base, wave, lane);
    for (int i = tid; i < FZ_L; i += 256) { tab[i] = a.b2[i]; tab[FZ_L + i] = a.gamma[i]; tab[2 * FZ_L + i] = a.beta[i]; tab[3 * FZ_L + i] = a.b1[i]; }

    const long long row = (long long)blockIdx.x * 64 + wave * 16 + l15;
    const bool live = row < a.rows;
    const long long rr = live ? row : a.rows - 1;
    v8 xh[KS], xl[KS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const float* p = a.src[s] + rr * a.ld[s] + 8 * g;
#pragma unroll
        for (int ks = 0; ks < FZ_KS; ++ks) {
            const float4 u0 = *reinterpret_cast<const float4*>(p + 32 * ks), u1 = *reinterpret_cast<const float4*>(p + 32 * ks + 4);
            const float v[8] = {u0.x, u0.y, u0.z, u0.w, u1.x, u1.y, u1.z, u1.w};
            uint4 o[2];
            split8<f16, 2>(v, o);
            xh[s * FZ_KS + ks] = as_v8<f16>(o[0]);
            xl[s * FZ_KS + ks] = as_v8<f16>(o[1]);
        }
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) { asm volatile("" : "+v"(xh[ks])); asm volatile("" : "+v"(xl[ks])); }

    uint4 hh[NCH], hl[NCH];
    f32x4 hacc[2];
#pragma unroll
    for (int sidx = 0; sidx < NST; ++sidx) {
        const int j = sidx / NS;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                               // stage sidx landed; every wave is done with the other stage
        if (sidx + 1 < NST) fz_dma64k(a.w1f + (long long)(sidx + 1) * (FZ_STAGE / 2), lds_base + ((sidx + 1) & 1) * FZ_STAGE, wave, lane);
        else fz_dma64k(a.w2f, lds_base + (NST & 1) * FZ_STAGE, wave, lane);
        if (NS == 1 || (sidx & 1) == 0) {
            const float4 b0 = *reinterpret_cast<const float4*>(tab + 3 * FZ_L + 32 * j + 4 * g), b1 = *reinterpret_cast<const float4*>(tab + 3 * FZ_L + 32 * j + 16 + 4 * g);
            hacc[0] = f32x4{b0.x, b0.y, b0.z, b0.w};
            hacc[1] = f32x4{b1.x, b1.y, b1.z, b1.w};
        }
        fz_stream<32, RD>(lrd + (sidx & 1) * FZ_STAGE, [&](int p, const uint4& wh, const uint4& wl) {
            const int n = NS == 1 ? (p >> 4) : (sidx & 1), ks = NS == 1 ? (p & 15) : p;
            hacc[n] = OpT<f16>::mfma(as_v8<f16>(wl), xh[ks], hacc[n]);
            hacc[n] = OpT<f16>::mfma(as_v8<f16>(wh), xl[ks], hacc[n]);
            hacc[n] = OpT<f16>::mfma(as_v8<f16>(wh), xh[ks], hacc[n]);
        });
        if (NS == 1 || (sidx & 1) == 1) {
            float v[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) { v[r] = swish_f(hacc[0][r]); v[4 + r] = swish_f(hacc[1][r]); }
            uint4 o[2];
            split8<f16, 2>(v, o);
            hh[j] = o[0]; hl[j] = o[1];
        }
    }

    f32x4 yacc[CF];
#pragma unroll
    for (int c = 0; c < CF; ++c) yacc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int stg = (NST + j) & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (j + 1 < NCH) fz_dma64k(a.w2f + (long long)(j + 1) * (FZ_STAGE / 2), lds_base + (stg ^ 1) * FZ_STAGE, wave, lane);
        fz_stream<CF, RD>(lrd + stg * FZ_STAGE, [&](int c, const uint4& wh, const uint4& wl) {
            yacc[c] = OpT<f16>::mfma(as_v8<f16>(wl), as_v8<f16>(hh[j]), yacc[c]);
            yacc[c] = OpT<f16>::mfma(as_v8<f16>(wh), as_v8<f16>(hl[j]), yacc[c]);
            yacc[c] = OpT<f16>::mfma(as_v8<f16>(wh), as_v8<f16>(hh[j]), yacc[c]);
        });
    }

    // epilogue: all loads before the first store (epilogues.h: stores share the VMEM counter with loads)
    float4 r0[FZ_KS], r1[FZ_KS];
    if (a.res != nullptr) {
        const float* p = a.res + rr * a.ld_res + 8 * g;
#pragma unroll
        for (int bp = 0; bp < FZ_KS; ++bp) { r0[bp] = *reinterpret_cast<const float4*>(p + 32 * bp); r1[bp] = *reinterpret_cast<const float4*>(p + 32 * bp + 4); }
    } else {
#pragma unroll
        for (int bp = 0; bp < FZ_KS; ++bp) { r0[bp] = make_float4(0.f, 0.f, 0.f, 0.f); r1[bp] = r0[bp]; }
    }
    fz_layer_norm(yacc, tab, g, a.eps);
    if (live) {
        float* dst = a.out + row * a.ld_out + 8 * g;
#pragma unroll
        for (int bp = 0; bp < FZ_KS; ++bp) {
            const f32x4 &x = yacc[2 * bp], &z = yacc[2 * bp + 1];
            *reinterpret_cast<float4*>(dst + 32 * bp) = make_float4(r0[bp].x + x[0], r0[bp].y + x[1], r0[bp].z + x[2], r0[bp].w + x[3]);
            *reinterpret_cast<float4*>(dst + 32 * bp + 4) = make_float4(r1[bp].x + z[0], r1[bp].y + z[1], r1[bp].z + z[2], r1[bp].w + z[3]);
        }
    }
}

}  // namespace skp

using namespace skp;

template <bool FC1, int NT>
static int launch_edge(const EdgeArgs& a, long long tiles, hipStream_t st) {
    auto kern = edge_update_kernel<FC1, NT>;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, FZ_SMEM) != hipSuccess) return SKGC_E_HIP;
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), FZ_SMEM, st, a);
    return hipGetLastError() == hipSuccess ? 0 : SKGC_E_HIP;
}

extern "C" {

int skgc_edge_update(const skgc_edge_desc* d, void* stream) {
    if (!d || !d->e_in || !d->recv || !d->w2f || !d->b2 || !d->gamma || !d->beta || !d->agg || d->rows <= 0 || (d->rows % FZ_TILE) || d->n_term < 0 || d->n_term > 2 ||
        (d->has_fc1 && !d->w1f) || (!d->has_fc1 && d->e_out) || (reinterpret_cast<size_t>(d->e_in) & 15) || (reinterpret_cast<size_t>(d->e_out) & 15))
        return SKGC_E_ARG;
    const long long tiles = d->rows / FZ_TILE;
    if (tiles > 0x7fffffff) return SKGC_E_ARG;
    EdgeArgs a;
    a.e_in = static_cast<const f16*>(d->e_in); a.e_out = static_cast<f16*>(d->e_out);
    for (int s = 0; s < 2; ++s) {
        a.term[s] = nullptr; a.idx[s] = nullptr; a.ld[s] = 0;
        if (s < d->n_term) {
            if (!d->term[s] || !d->idx[s] || d->ld[s] < FZ_L || (d->ld[s] & 3) || (reinterpret_cast<size_t>(d->term[s]) & 15)) return SKGC_E_ARG;
            a.term[s] = d->term[s]; a.idx[s] = d->idx[s]; a.ld[s] = d->ld[s];
        }
    }
    a.recv = d->recv; a.w1f = static_cast<const f16*>(d->w1f); a.w2f = static_cast<const f16*>(d->w2f);
    a.b2 = d->b2; a.gamma = d->gamma; a.beta = d->beta; a.agg = d->agg; a.heads = d->heads; a.eps = 1e-5f;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (d->has_fc1) {
        if (d->n_term == 2) return launch_edge<true, 2>(a, tiles, st);
        if (d->n_term == 1) return launch_edge<true, 1>(a, tiles, st);
        return launch_edge<true, 0>(a, tiles, st);
    }
    if (d->n_term == 2) return launch_edge<false, 2>(a, tiles, st);
    if (d->n_term == 1) return launch_edge<false, 1>(a, tiles, st);
    return launch_edge<false, 0>(a, tiles, st);
}

int skgc_node_mlp(const skgc_node_desc* d, void* stream) {
    if (!d || !d->w1f || !d->w2f || !d->b1 || !d->b2 || !d->gamma || !d->beta || !d->out || d->rows <= 0 || d->n_src < 1 || d->n_src > 2 || d->ld_out < FZ_L || (d->ld_out & 3) ||
        (d->res && (d->ld_res < FZ_L || (d->ld_res & 3))) || (reinterpret_cast<size_t>(d->out) & 15) || (reinterpret_cast<size_t>(d->res) & 15))
        return SKGC_E_ARG;
    NodeArgs a;
    for (int s = 0; s < 2; ++s) {
        a.src[s] = nullptr; a.ld[s] = 0;
        if (s < d->n_src) {
            if (!d->src[s] || d->ld[s] < FZ_L || (d->ld[s] & 3) || (reinterpret_cast<size_t>(d->src[s]) & 15)) return SKGC_E_ARG;
            a.src[s] = d->src[s]; a.ld[s] = d->ld[s];
        }
    }
    a.w1f = static_cast<const f16*>(d->w1f); a.w2f = static_cast<const f16*>(d->w2f);
    a.b1 = d->b1; a.b2 = d->b2; a.gamma = d->gamma; a.beta = d->beta;
    a.res = d->res; a.ld_res = d->ld_res; a.out = d->out; a.ld_out = d->ld_out; a.rows = d->rows; a.eps = 1e-5f;
    const long long tiles = (d->rows + 63) / 64;
    if (tiles > 0x7fffffff) return SKGC_E_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    auto go = [&](auto kern) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, FZ_SMEM) != hipSuccess) return SKGC_E_HIP;
        hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), FZ_SMEM, st, a);
        return hipGetLastError() == hipSuccess ? 0 : SKGC_E_HIP;
    };
    return d->n_src == 2 ? go(node_mlp_kernel<2>) : go(node_mlp_kernel<1>);
}

int skgc_segment_fixup(float* agg, const float* heads, const int* nodes, const int* first, const int* tiles, int n_nodes, void* stream) {
    if (n_nodes == 0) return 0;
    if (!agg || !heads || !nodes || !first || !tiles || n_nodes < 0) return SKGC_E_ARG;
    hipLaunchKernelGGL(segment_fixup_kernel, dim3((unsigned)n_nodes), dim3(128), 0, static_cast<hipStream_t>(stream), agg, heads, nodes, first, tiles);
    return hipGetLastError() == hipSuccess ? 0 : SKGC_E_HIP;
}

}  // extern "C"
