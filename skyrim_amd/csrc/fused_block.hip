// Everything of an EarthSpecificBlock that follows the attention, as ONE kernel, in place on the residual planes:
//
//     x[t] <- x_mid + LayerNorm(norm2)( fc2( GELU( fc1( x_mid ) ) ) ),      x_mid = x[t] + LayerNorm(norm1)( ao[win(t)] Wp^T + b )
//
// t runs over the stream's tokens, win(t) = the inverse of the window table (window reverse + un-roll + crop as a GATHER of the
// attention output's rows: the stream itself is read and written in whole 1 KiB blocks, and the tile count is that of the tokens --
// 1024 tiles = 4 full rounds of the 256 CUs at 91 x 180 x 8, where the 5.5 % of padded window rows would make it 1080 = 5 rounds).
// It is rowtile.hip's projection followed by fused_mlp.hip's MLP on the same register-resident rows:
//
//   phase 1  projection: the wave's FM x 16 attention rows are MFMA B-operand fragments (hi / lo planes), the weights stream through
//            LDS in blocks of 32 output columns (two stages, one barrier per block), the C outputs of a row end up in its lane quad's
//            accumulators.
//   between  + bias, LayerNorm (in-lane sums + two shuffles), + the gathered residual row: with the perm8 row order of the prepared
//            weights the accumulators of fragment pair bp are columns 32 bp + 8 (lane >> 4) + [0..7] -- after the hi / lo split they
//            ARE input fragment bp of the MLP.  x_mid never exists in HBM.
//   phase 2  MLP exactly as fused_mlp.hip (hidden in 32-unit chunks, fc1 block in stage 0, fc2 block in stage 1 -- the projection's
//            two stages), epilogue + fc2 bias, LayerNorm, + x_mid from registers, stored as whole 1 KiB blocks.
//
// Against the two kernels it replaces: the stream is read once and written once per block instead of twice (8 of 20 bytes per
// element), one launch less, and the projection's MFMAs run inside a compute-bound kernel instead of an HBM-bound one.
// gfx950 only.
#include <cstdlib>
#include "gemm_dma.h"
#include "launchers.h"

namespace skp {

template <int C_, int FM_, int NWAVES_, int WPE_>
struct BlockShape {
    static constexpr int C = C_, FM = FM_, NWAVES = NWAVES_, THREADS = 64 * NWAVES_, WPE = WPE_;
    static constexpr int KS = C / 32, CF = C / 16, HID = 4 * C, NCH = HID / 32, BM = NWAVES * FM * 16;
    static constexpr int NPB = C / 32;                // projection: blocks of 32 output columns
    static constexpr int P_BLK = KS * 2 * 2;          // KiB per projection block / per fc1 chunk: [ks][n][plane]
    static constexpr int W2_BLK = CF * 2;             // KiB per fc2 chunk: [c][plane]
    static constexpr int STAGE = P_BLK * 1024;
    static constexpr int T_PB = 0, T_G1 = C, T_E1 = 2 * C, T_B1 = 3 * C, T_B2 = 3 * C + HID, T_G2 = T_B2 + C, T_E2 = T_G2 + C;
    static constexpr int TAB = T_E2 + C;
    static constexpr int SMEM = 2 * STAGE + TAB * 4;
    static_assert(P_BLK % NWAVES == 0 && W2_BLK % NWAVES == 0 && W2_BLK == P_BLK && NPB % 2 == 0, "stages: equal size, even block count, whole DMA blocks per wave");
    static_assert(SMEM <= 160 * 1024, "LDS");
};

template <class T>
struct BlockArgs {
    const T* ao; long long ao_plane;       // attention output, window-ordered rows, blocked layout [rows/16][C/32][16][32]
    int M;                                 // stream tokens (multiple of 16)
    T* xs; long long xs_plane;             // residual stream planes, blocked layout
    const int* winv;                       // stream token -> window row of the attention output
    const T *projf, *w1f, *w2f;            // fragment-order weights (prep_rowtile_weights, prep_mlp_weights)
    const float *proj_b, *g1, *e1, *b1, *b2, *g2, *e2;
    float eps;
};

__device__ __forceinline__ void fb_ld_pair(const char* p, uint4 (&w)[2]) {
    w[0] = *reinterpret_cast<const uint4*>(p);
    w[1] = *reinterpret_cast<const uint4*>(p + 1024);
    __builtin_amdgcn_sched_barrier(0);          // keep the reads HERE, ahead of the MFMAs that follow (fused_mlp.hip)
}

template <class T, class S>
__global__ void __launch_bounds__(S::THREADS) __attribute__((amdgpu_waves_per_eu(S::WPE, S::WPE)))
proj_mlp_kernel(const BlockArgs<T> a) {
    constexpr int C = S::C, FM = S::FM, KS = S::KS, CF = S::CF, NCH = S::NCH, NPB = S::NPB, NWAVES = S::NWAVES, DEPTH = 3, NS = KS * 2;
    typedef typename OpT<T>::v8 v8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* st0 = smem;
    char* st1 = smem + S::STAGE;
    float* tab = reinterpret_cast<float*>(smem + 2 * S::STAGE);
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds_base = (unsigned)(size_t)smem;

    auto dma = [&](const T* src_blocks, unsigned dst) {             // P_BLK consecutive KiB -> one stage, block b by wave b % NWAVES
        const T* src = src_blocks + lane * 8;
#pragma unroll
        for (int i = 0; i < S::P_BLK / NWAVES; ++i) {
            const int b = wave + i * NWAVES;
            glds16(src + (b << 9), dst + (unsigned)(b << 10));
        }
    };
    dma(a.projf, lds_base);

    for (int i = tid; i < C; i += S::THREADS) {
        tab[S::T_PB + i] = a.proj_b[i]; tab[S::T_G1 + i] = a.g1[i]; tab[S::T_E1 + i] = a.e1[i];
        tab[S::T_B2 + i] = a.b2[i]; tab[S::T_G2 + i] = a.g2[i]; tab[S::T_E2 + i] = a.e2[i];
    }
    for (int i = tid; i < S::HID; i += S::THREADS) tab[S::T_B1 + i] = a.b1[i];

    // the attention rows of the wave's tokens as B-operand fragments (gathered: 16 bytes per lane, 64 bytes per row and k-step)
    const long long rb0 = (long long)blockIdx.x * (S::BM / 16) + wave * FM;
    v8 xh[FM][KS], xl[FM][KS];
    bool live[FM];
#pragma unroll
    for (int t = 0; t < FM; ++t) {
        live[t] = (rb0 + t) * 16 < a.M;
        const int src = live[t] ? a.winv[(rb0 + t) * 16 + l15] : 0;
        const T* p = a.ao + blk_off(src, g * 8, C);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            xh[t][ks] = *reinterpret_cast<const v8*>(p + (ks << 9));
            xl[t][ks] = *reinterpret_cast<const v8*>(p + (ks << 9) + a.ao_plane);
        }
    }
#pragma unroll
    for (int t = 0; t < FM; ++t) {              // consumed once here: hipcc's vmcnt waits for these loads sit BEFORE the loops (fused_mlp.hip)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) { asm volatile("" : "+v"(xh[t][ks])); asm volatile("" : "+v"(xl[t][ks])); }
    }

    f32x4 yacc[FM][CF];
#pragma unroll
    for (int t = 0; t < FM; ++t)
#pragma unroll
        for (int c = 0; c < CF; ++c) yacc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    // ---- phase 1: projection, blocks of 32 output columns alternating between the two stages ------------------------------- //
#pragma unroll
    for (int j = 0; j < NPB; ++j) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                               // block j landed; every wave is done with block j - 1
        if (j + 1 < NPB) dma(a.projf + ((long long)(j + 1) * S::P_BLK << 9), lds_base + (unsigned)(((j + 1) & 1) * S::STAGE));
        else dma(a.w1f, lds_base);                     // NPB is even: the last block sits in stage 1, stage 0 is free for fc1's chunk 0
        const char* st = (j & 1) ? st1 : st0;
        uint4 ring[DEPTH][2];
#pragma unroll
        for (int s = 0; s < DEPTH - 1 && s < NS; ++s) fb_ld_pair(st + ((s * 2) << 10) + lane * 16, ring[s % DEPTH]);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (s + DEPTH - 1 < NS) fb_ld_pair(st + (((s + DEPTH - 1) * 2) << 10) + lane * 16, ring[(s + DEPTH - 1) % DEPTH]);
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

    // ---- between: x_mid = x + LayerNorm(yacc + bias) -> the MLP's input fragments (the attention fragments are dead) ----------- //
#pragma unroll
    for (int t = 0; t < FM; ++t) {
        const T* old = a.xs + ((live[t] ? rb0 + t : 0) * KS << 9) + l15 * 32 + g * 8;
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
            const float4 b0 = *reinterpret_cast<const float4*>(tab + S::T_PB + n), b1 = *reinterpret_cast<const float4*>(tab + S::T_PB + n + 4);
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
#pragma unroll
        for (int bp = 0; bp < KS; ++bp) {
            const int n = 32 * bp + 8 * g;
            const float4 g0 = *reinterpret_cast<const float4*>(tab + S::T_G1 + n), g1 = *reinterpret_cast<const float4*>(tab + S::T_G1 + n + 4);
            const float4 e0 = *reinterpret_cast<const float4*>(tab + S::T_E1 + n), e1 = *reinterpret_cast<const float4*>(tab + S::T_E1 + n + 4);
            const f32x4 &x = yacc[t][2 * bp], &z = yacc[t][2 * bp + 1];
            const float y[8] = {(x[0] - mean) * rstd * g0.x + e0.x, (x[1] - mean) * rstd * g0.y + e0.y, (x[2] - mean) * rstd * g0.z + e0.z, (x[3] - mean) * rstd * g0.w + e0.w,
                                (z[0] - mean) * rstd * g1.x + e1.x, (z[1] - mean) * rstd * g1.y + e1.y, (z[2] - mean) * rstd * g1.z + e1.z, (z[3] - mean) * rstd * g1.w + e1.w};
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = ((float)oh[bp][i] + (float)ol[bp][i]) + y[i];
            uint4 o[2];
            split8<T, 2>(v, o);
            xh[t][bp] = as_v8<T>(o[0]);
            xl[t][bp] = as_v8<T>(o[1]);
        }
#pragma unroll
        for (int c = 0; c < CF; ++c) yacc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // ---- phase 2: the MLP of fused_mlp.hip on x_mid (plain schedule: fc1(j) | GELU(j) | fc2(j), two barriers per chunk) ---------- //
    const unsigned ldsA = lds_base, ldsB = lds_base + S::STAGE;
    for (int j = 0; j < NCH; ++j) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                               // fc1 block j landed; every wave is done with fc2 block j - 1 (j = 0: with the projection)
        {
            const T* src = a.w2f + ((long long)j * S::W2_BLK << 9) + lane * 8;
#pragma unroll
            for (int i = 0; i < S::W2_BLK / NWAVES; ++i) {
                const int b = wave + i * NWAVES;
                glds16(src + (b << 9), ldsB + (unsigned)(b << 10));
            }
        }
        f32x4 hacc[FM][2];
#pragma unroll
        for (int t = 0; t < FM; ++t) { hacc[t][0] = f32x4{0.f, 0.f, 0.f, 0.f}; hacc[t][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        {
            uint4 ring[DEPTH][2];
#pragma unroll
            for (int s = 0; s < DEPTH - 1 && s < NS; ++s) fb_ld_pair(st0 + ((s * 2) << 10) + lane * 16, ring[s % DEPTH]);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                if (s + DEPTH - 1 < NS) fb_ld_pair(st0 + (((s + DEPTH - 1) * 2) << 10) + lane * 16, ring[(s + DEPTH - 1) % DEPTH]);
                const int ks = s >> 1, n = s & 1;
                const uint4 wh = ring[s % DEPTH][0], wl = ring[s % DEPTH][1];
#pragma unroll
                for (int t = 0; t < FM; ++t) hacc[t][n] = OpT<T>::mfma(as_v8<T>(wl), xh[t][ks], hacc[t][n]);
#pragma unroll
                for (int t = 0; t < FM; ++t) hacc[t][n] = OpT<T>::mfma(as_v8<T>(wh), xl[t][ks], hacc[t][n]);
#pragma unroll
                for (int t = 0; t < FM; ++t) hacc[t][n] = OpT<T>::mfma(as_v8<T>(wh), xh[t][ks], hacc[t][n]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        uint4 hh[FM], hl[FM];
        {
            const float4 bb0 = *reinterpret_cast<const float4*>(tab + S::T_B1 + j * 32 + 4 * g), bb1 = *reinterpret_cast<const float4*>(tab + S::T_B1 + j * 32 + 16 + 4 * g);
#pragma unroll
            for (int t = 0; t < FM; ++t) {
                const f32x2 a0 = gelu_erf2(f32x2{hacc[t][0][0] + bb0.x, hacc[t][0][1] + bb0.y}), a1 = gelu_erf2(f32x2{hacc[t][0][2] + bb0.z, hacc[t][0][3] + bb0.w});
                const f32x2 a2 = gelu_erf2(f32x2{hacc[t][1][0] + bb1.x, hacc[t][1][1] + bb1.y}), a3 = gelu_erf2(f32x2{hacc[t][1][2] + bb1.z, hacc[t][1][3] + bb1.w});
                const float v[8] = {a0.x, a0.y, a1.x, a1.y, a2.x, a2.y, a3.x, a3.y};
                uint4 o[2];
                split8<T, 2>(v, o);
                hh[t] = o[0]; hl[t] = o[1];
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                               // fc2 block j landed; every wave is done with fc1 block j
        if (j + 1 < NCH) dma(a.w1f + ((long long)(j + 1) * S::P_BLK << 9), ldsA);
        {
            uint4 ring[DEPTH][2];
#pragma unroll
            for (int c = 0; c < DEPTH - 1 && c < CF; ++c) fb_ld_pair(st1 + ((c * 2) << 10) + lane * 16, ring[c % DEPTH]);
#pragma unroll
            for (int c = 0; c < CF; ++c) {
                if (c + DEPTH - 1 < CF) fb_ld_pair(st1 + (((c + DEPTH - 1) * 2) << 10) + lane * 16, ring[(c + DEPTH - 1) % DEPTH]);
                const uint4 wh = ring[c % DEPTH][0], wl = ring[c % DEPTH][1];
#pragma unroll
                for (int t = 0; t < FM; ++t) yacc[t][c] = OpT<T>::mfma(as_v8<T>(wl), as_v8<T>(hh[t]), yacc[t][c]);
#pragma unroll
                for (int t = 0; t < FM; ++t) yacc[t][c] = OpT<T>::mfma(as_v8<T>(wh), as_v8<T>(hl[t]), yacc[t][c]);
#pragma unroll
                for (int t = 0; t < FM; ++t) yacc[t][c] = OpT<T>::mfma(as_v8<T>(wh), as_v8<T>(hh[t]), yacc[t][c]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    // ---- epilogue: + fc2 bias, LayerNorm(norm2), + x_mid (registers), whole blocks of the stream ------------------------------------ //
#pragma unroll
    for (int t = 0; t < FM; ++t) {
        float s = 0.f;
#pragma unroll
        for (int bp = 0; bp < KS; ++bp) {
            const int n = 32 * bp + 8 * g;
            const float4 b0 = *reinterpret_cast<const float4*>(tab + S::T_B2 + n), b1 = *reinterpret_cast<const float4*>(tab + S::T_B2 + n + 4);
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
        if (!live[t]) continue;
        T* dst = a.xs + ((rb0 + t) * KS << 9) + l15 * 32 + g * 8;
#pragma unroll
        for (int bp = 0; bp < KS; ++bp) {
            const int n = 32 * bp + 8 * g;
            const float4 g0 = *reinterpret_cast<const float4*>(tab + S::T_G2 + n), g1 = *reinterpret_cast<const float4*>(tab + S::T_G2 + n + 4);
            const float4 e0 = *reinterpret_cast<const float4*>(tab + S::T_E2 + n), e1 = *reinterpret_cast<const float4*>(tab + S::T_E2 + n + 4);
            float oh[8], ol[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { oh[i] = (float)xh[t][bp][i]; ol[i] = (float)xl[t][bp][i]; }
            const f32x4 &x = yacc[t][2 * bp], &z = yacc[t][2 * bp + 1];
            const float v[8] = {(oh[0] + ol[0]) + ((x[0] - mean) * rstd * g0.x + e0.x), (oh[1] + ol[1]) + ((x[1] - mean) * rstd * g0.y + e0.y),
                                (oh[2] + ol[2]) + ((x[2] - mean) * rstd * g0.z + e0.z), (oh[3] + ol[3]) + ((x[3] - mean) * rstd * g0.w + e0.w),
                                (oh[4] + ol[4]) + ((z[0] - mean) * rstd * g1.x + e1.x), (oh[5] + ol[5]) + ((z[1] - mean) * rstd * g1.y + e1.y),
                                (oh[6] + ol[6]) + ((z[2] - mean) * rstd * g1.z + e1.z), (oh[7] + ol[7]) + ((z[3] - mean) * rstd * g1.w + e1.w)};
            store8_planes<T, 2>(dst + (bp << 9), a.xs_plane, v);
        }
    }
}

template <class T, class S>
static hipError_t launch_block(const BlockArgs<T>& a, hipStream_t s) {
    auto kern = proj_mlp_kernel<T, S>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, S::SMEM);
    if (e != hipSuccess) return e;
    const unsigned grid = (unsigned)((a.M + S::BM - 1) / S::BM);
    if (grid == 0) return hipSuccess;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(S::THREADS), S::SMEM, s, a);
    return hipGetLastError();
}

template <class P>
hipError_t op_proj_mlp_fused(const Geom& g, const BlockW<typename P::T>& b, const int* winv, int res, typename P::T* Xs, const Work<P>& wk, hipStream_t s) {
    typedef typename P::T T;
    static_assert(P::NA == 2 && P::NW == 2, "3-term path");
    BlockArgs<T> a{wk.ao, wk.ao_plane, g.ntok[res], Xs, wk.xs_plane[res], winv, b.projf, b.w1f, b.w2f,
                   b.proj_b, b.n1_g, b.n1_b, b.fc1_b, b.fc2_b, b.n2_g, b.n2_b, 1e-5f};
    if (a.M % 16 != 0) return hipErrorInvalidValue;
    // the shapes fused_mlp.hip settled on: C = 192 two 4-wave workgroups per CU with 32 rows per wave, C = 384 one 8-wave workgroup
    if (res == 0) return launch_block<T, BlockShape<192, 2, 4, 2>>(a, s);
    return launch_block<T, BlockShape<384, 1, 8, 2>>(a, s);
}
template hipError_t op_proj_mlp_fused<PrecBF16x3>(const Geom&, const BlockW<bf16>&, const int*, int, bf16*, const Work<PrecBF16x3>&, hipStream_t);
template hipError_t op_proj_mlp_fused<PrecF16x3>(const Geom&, const BlockW<f16>&, const int*, int, f16*, const Work<PrecF16x3>&, hipStream_t);

}  // namespace skp
