// Everything of an EarthSpecificBlock that follows the attention as ONE kernel (fused_block.hip), in the TWO-TERM form
//
//     A W^T  ~  A_hi W_hi^T + A_lo W_hi^T          (activations stay hi/lo pairs, the weights are ONE fp16 plane)
//
// Why two terms.  The prepared weights are constants; rounding them to 11 significant bits perturbs the model by 2^-12 relative per
// weight (measured on the oracle with exactly this plan -- proj, fc1, fc2 weights rounded, QKV activation rounded: 4.2e-4 per-channel
// error after one step, 5.3e-4 after four, bar 1e-3).  It removes a third of the MFMAs, HALF of the bytes that cross LDS
// (the token tile lives in registers, only weights stream through LDS) and half of the LDS-DMA traffic.
//
// Launch shape: 4-wave workgroups, TWO per CU (one wave of each on every SIMD), two LDS slots of one chunk each and two barriers per
// 32-unit chunk:  | fc1(j) | GELU(j) | fc2(j) |.  The hi-only chunk is 24 KB at C = 384, so two workgroups fit a CU where the three-term
// kernel (fused_block.hip) fits one; the two run free of each other and drift out of phase, so that one's GELU / LayerNorm / row I/O runs
// under the other's MFMAs.  What else was measured -- one 8-wave workgroup with a five-slot ring, its halves half a chunk apart, four
// accumulator chains, a one-wave-per-SIMD form with 32 / 64 rows per wave, software pipelining inside the wave, and the per-phase timing
// probes -- is in docs/experiments.md; none of it is in the tree.
//
// ONE (term-plan bits 8-11): the ACTIVATION operands too as one fp16 plane -- A_hi W only: attention output, mid-block stream and hidden
// activation are rounded to fp16 where they enter a GEMM (the residual path keeps the hi/lo pair), the attention output's lo plane is
// neither written nor read.  Half of the two-term form's MFMAs; meant for weights rounded with error feedback against these very operands
// (pangu/calibration.py), which is what pays for the activation rounding.
//
// Row-tile structure, epilogues and data layouts are fused_block.hip's: a wave owns FM x 16 stream tokens, their attention rows are
// gathered through the inverse window table as MFMA B-operand fragments, x_mid = x + LayerNorm(proj) becomes the MLP's operand in
// registers (perm8), the stream is read once and written once.  gfx950 only.
#include "blockrow.h"
#include "launchers.h"

namespace skp {

template <int C_, int FM_, bool ONE_>
struct Blk2Shape {
    static constexpr bool ONE = ONE_;
    static constexpr int C = C_, FM = FM_, NWAVES = 4, THREADS = 64 * NWAVES, RD = 2;
    static constexpr int KS = C / 32, CF = C / 16, HID = 4 * C, NCH = HID / 32, NPB = C / 32, BM = NWAVES * FM * 16;
    static constexpr int SLOT_KIB = KS * 2;          // KiB of a projection block [ks][n] = of an fc1 chunk [ks][n] = of an fc2 chunk [c]
    static constexpr int SLOT = SLOT_KIB * 1024, NSLOT = 2;
    static constexpr int T_PB = 0, T_G1 = C, T_E1 = 2 * C, T_B1 = 3 * C, T_B2 = 3 * C + HID, T_G2 = T_B2 + C, T_E2 = T_G2 + C;
    static constexpr int TAB = T_E2 + C;
    static constexpr int SMEM = NSLOT * SLOT + TAB * 4;
    static_assert(SLOT_KIB % NWAVES == 0 && NPB % 2 == 0 && NCH >= 3, "DMA pieces per wave; even projection block count");
    static_assert(SMEM <= 80 * 1024, "two workgroups per CU");
};

template <class T, class S>
__global__ void __launch_bounds__(S::THREADS) __attribute__((amdgpu_waves_per_eu(2, 2)))
proj_mlp2_kernel(const Block2Args<T> a) {
    constexpr int C = S::C, FM = S::FM, KS = S::KS, CF = S::CF, NCH = S::NCH, NPB = S::NPB, NWAVES = S::NWAVES, RD = S::RD;
    typedef typename OpT<T>::v8 v8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* tab = reinterpret_cast<float*>(smem + S::NSLOT * S::SLOT);
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds_base = (unsigned)(size_t)smem;
    const char* lrd = smem + lane * 16;                          // a lane's 16 bytes of every fragment

    // SLOT_KIB consecutive KiB of `src` -> LDS `dst`, piece q by wave q % NWAVES
    auto dma1 = [&](const T* src, unsigned dst) {
        const T* s = src + lane * 8;
#pragma unroll
        for (int i = 0; i < S::SLOT_KIB / NWAVES; ++i) {
            const int q = wave + i * NWAVES;
            glds16(s + (q << 9), dst + (unsigned)(q << 10));
        }
    };
    dma1(a.projh, lds_base);

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
            if constexpr (S::ONE) xl[t][ks] = v8{};              // the lo plane of the attention output is not read at all
            else xl[t][ks] = *reinterpret_cast<const v8*>(p + (ks << 9) + a.ao_plane);
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

    // ---- phase 1: projection, blocks of 32 output columns alternating between the two slots -------------------------------------- //
#pragma unroll
    for (int j = 0; j < NPB; ++j) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                               // block j landed; every wave is done with block j - 1
        if (j + 1 < NPB) dma1(a.projh + ((long long)(j + 1) * S::SLOT_KIB << 9), lds_base + (unsigned)(((j + 1) & 1) * S::SLOT));
        else dma1(a.w1h, lds_base);                    // NPB is even: the last block sits in slot 1, slot 0 is free for fc1's chunk 0
        sk_stream<KS, RD>(lrd + (j & 1) * S::SLOT, [&](int ks, const uint4& w0, const uint4& w1) {
            if constexpr (!S::ONE) {
#pragma unroll
                for (int t = 0; t < FM; ++t) yacc[t][2 * j] = OpT<T>::mfma(as_v8<T>(w0), xl[t][ks], yacc[t][2 * j]);
#pragma unroll
                for (int t = 0; t < FM; ++t) yacc[t][2 * j + 1] = OpT<T>::mfma(as_v8<T>(w1), xl[t][ks], yacc[t][2 * j + 1]);
            }
#pragma unroll
            for (int t = 0; t < FM; ++t) yacc[t][2 * j] = OpT<T>::mfma(as_v8<T>(w0), xh[t][ks], yacc[t][2 * j]);
#pragma unroll
            for (int t = 0; t < FM; ++t) yacc[t][2 * j + 1] = OpT<T>::mfma(as_v8<T>(w1), xh[t][ks], yacc[t][2 * j + 1]);
        });
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

    // ---- phase 2: the MLP, W1(j) in slot 0 and W2(j) in slot 1 ---------------------------------------------------------------------- //
    f32x4 hacc[FM][2];
    uint4 hh[FM], hl[FM];
    for (int j = 0; j < NCH; ++j) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                               // W1(j) landed in slot 0; every wave is done with W2(j - 1) / the projection in slot 1
        dma1(a.w2h + ((long long)j * S::SLOT_KIB << 9), lds_base + S::SLOT);
#pragma unroll
        for (int t = 0; t < FM; ++t) { hacc[t][0] = f32x4{0.f, 0.f, 0.f, 0.f}; hacc[t][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        sk_stream<KS, RD>(lrd, [&](int ks, const uint4& w0, const uint4& w1) {
            if constexpr (!S::ONE) {
#pragma unroll
                for (int t = 0; t < FM; ++t) hacc[t][0] = OpT<T>::mfma(as_v8<T>(w0), xl[t][ks], hacc[t][0]);
#pragma unroll
                for (int t = 0; t < FM; ++t) hacc[t][1] = OpT<T>::mfma(as_v8<T>(w1), xl[t][ks], hacc[t][1]);
            }
#pragma unroll
            for (int t = 0; t < FM; ++t) hacc[t][0] = OpT<T>::mfma(as_v8<T>(w0), xh[t][ks], hacc[t][0]);
#pragma unroll
            for (int t = 0; t < FM; ++t) hacc[t][1] = OpT<T>::mfma(as_v8<T>(w1), xh[t][ks], hacc[t][1]);
        });
        {   // bias + GELU + hi/lo split: the lane's 8 hidden units 16 n + 4 g + r become k-slots 8 g + 4 n + r of fc2
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
        __syncthreads();                               // W2(j) landed; every wave is done with W1(j)
        if (j + 1 < NCH) dma1(a.w1h + ((long long)(j + 1) * S::SLOT_KIB << 9), lds_base);
        sk_stream<CF / 2, RD>(lrd + S::SLOT, [&](int p, const uint4& w0, const uint4& w1) {
            if constexpr (!S::ONE) {
#pragma unroll
                for (int t = 0; t < FM; ++t) yacc[t][2 * p] = OpT<T>::mfma(as_v8<T>(w0), as_v8<T>(hl[t]), yacc[t][2 * p]);
#pragma unroll
                for (int t = 0; t < FM; ++t) yacc[t][2 * p + 1] = OpT<T>::mfma(as_v8<T>(w1), as_v8<T>(hl[t]), yacc[t][2 * p + 1]);
            }
#pragma unroll
            for (int t = 0; t < FM; ++t) yacc[t][2 * p] = OpT<T>::mfma(as_v8<T>(w0), as_v8<T>(hh[t]), yacc[t][2 * p]);
#pragma unroll
            for (int t = 0; t < FM; ++t) yacc[t][2 * p + 1] = OpT<T>::mfma(as_v8<T>(w1), as_v8<T>(hh[t]), yacc[t][2 * p + 1]);
        });
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
static hipError_t launch_blk2(const Block2Args<T>& a, hipStream_t s) {
    auto kern = proj_mlp2_kernel<T, S>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, S::SMEM);
    if (e != hipSuccess) return e;
    const unsigned grid = (unsigned)((a.M + S::BM - 1) / S::BM);
    if (grid == 0) return hipSuccess;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(S::THREADS), S::SMEM, s, a);
    return hipGetLastError();
}

// `one`: the layer's term-plan bit 8 + l -- the activation operands as one fp16 plane too
hipError_t op_proj_mlp2(const Geom& g, const BlockW<f16>& b, const int* winv, int res, f16* Xs, const Work<PrecF16x3>& wk, hipStream_t s, int one) {
    typedef f16 T;
    Block2Args<T> a{wk.ao, wk.ao_plane, g.ntok[res], Xs, wk.xs_plane[res], winv, b.projh, b.w1h, b.w2h,
                    b.proj_b, b.n1_g, b.n1_b, b.fc1_b, b.fc2_b, b.n2_g, b.n2_b, 1e-5f};
    if (a.M % 16 != 0) return hipErrorInvalidValue;
    if (res == 0) return one ? launch_blk2<T, Blk2Shape<192, 2, true>>(a, s) : launch_blk2<T, Blk2Shape<192, 2, false>>(a, s);
    return one ? launch_blk2<T, Blk2Shape<384, 1, true>>(a, s) : launch_blk2<T, Blk2Shape<384, 1, false>>(a, s);
}

}  // namespace skp
