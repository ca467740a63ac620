// Everything of an EarthSpecificBlock that follows the attention as ONE kernel (fused_block.hip), in the TWO-TERM form
//
//     A W^T  ~  A_hi W_hi^T + A_lo W_hi^T          (activations stay hi/lo pairs, the weights are ONE fp16 plane)
//
// with the MLP's chunk loop SOFTWARE-PIPELINED inside every wave.
//
// Why two terms.  The prepared weights are constants; rounding them to 11 significant bits perturbs the model by 2^-12 relative per
// weight (measured on the oracle with exactly this plan -- proj, fc1, fc2 weights rounded, QKV activation rounded: 4.2e-4 per-channel
// error after one step, 5.3e-4 after four, bar 1e-3).  It removes a third of the MFMAs, HALF of the bytes that cross LDS
// (the token tile lives in registers, only weights stream through LDS) and half of the LDS-DMA traffic.
//
// Why the pipelining.  With 256 registers a SIMD holds two waves, and the timing probes of this kernel's first form (fc1(j) | GELU(j) |
// fc2(j) per wave; PROBE below) showed every removed phase coming off the kernel time ONE FOR ONE -- GELU, the LDS reads' start-up
// latencies, the LDS-DMA issue, the row gathers and stores: a wave is a serial chain, two chains per SIMD do not fill the matrix pipe
// (45 % busy), and re-ordering work BETWEEN the waves (the halves of the workgroup half a chunk apart, two 4-wave workgroups per CU with
// a start stagger) changes nothing as long as each chain is that long.  What shortens a chain is overlap INSIDE the wave:
//
//     interval j:   [ fc1(j) MFMAs  +  GELU(j-1) spliced between them ]   [ fc2(j-1) MFMAs  +  the LDS-DMA requests of W1(j+1), W2(j) ]
//
// fc2 runs one chunk behind fc1, so the GELU of chunk j-1 (VALU, ~110 instructions per 16 tokens) has no consumer waiting for it and
// sits in the shadow of fc1(j)'s MFMAs; the weight-fragment ring runs through both phases without draining (one exposed LDS latency
// per interval instead of two); the DMA requests ride between fc2's MFMAs instead of in front of the interval.  One barrier per
// interval.  Ring: W1(j) in slot j & 1, W2(j) in slot 2 + (j & 1); the projection's 32-column blocks alternate between slots 2 and 3
// while W1(0) lands in slot 0.
//
// Row-tile structure, epilogues and data layouts are fused_block.hip's: a wave owns FM x 16 stream tokens, their attention rows are
// gathered through the inverse window table as MFMA B-operand fragments, x_mid = x + LayerNorm(proj) becomes the MLP's operand in
// registers (perm8), the stream is read once and written once.  gfx950 only.
#include <cstdlib>
#include "gemm_dma.h"
#include "launchers.h"

namespace skp {

template <int C_, int FM_, int RD_, int NV_, int PROBE_ = 0>
struct Blk2Shape {
    // timing probes (measurement only; results are wrong): 1 no GELU polynomial, 2 one fragment pair read per phase, 4 no weight DMA
    // in the MLP loop, 8 no barrier in the MLP loop, 16 no row gathers / stores
    static constexpr int PROBE = PROBE_;
    static constexpr int C = C_, FM = FM_, NWAVES = 8, THREADS = 512, RD = RD_;
    static constexpr int NV = NV_;                   // VALU instructions the scheduler is asked to place behind every MFMA of a spliced step (0: its own choice)
    static constexpr int KS = C / 32, CF = C / 16, HID = 4 * C, NCH = HID / 32, NPB = C / 32, BM = NWAVES * FM * 16;
    static constexpr int SLOT_KIB = KS * 2;          // KiB of a projection block [ks][n] = of an fc1 chunk [ks][n] = of an fc2 chunk [c]
    static constexpr int SLOT = SLOT_KIB * 1024, NSLOT = 4;
    static constexpr int PPW = 2 * SLOT_KIB / NWAVES;   // DMA pieces per wave and interval (one W1 chunk + one W2 chunk)
    static constexpr int T_PB = 0, T_G1 = C, T_E1 = 2 * C, T_B1 = 3 * C, T_B2 = 3 * C + HID, T_G2 = T_B2 + C, T_E2 = T_G2 + C;
    static constexpr int TAB = T_E2 + C;
    static constexpr int SMEM = NSLOT * SLOT + TAB * 4;
    static_assert((2 * SLOT_KIB) % NWAVES == 0 && NPB % 2 == 0 && NCH >= 3 && PPW <= KS, "DMA pieces per wave; even projection block count");
    static_assert(SMEM <= 160 * 1024, "LDS");
};

template <class T>
struct Block2Args {
    const T* ao; long long ao_plane;       // attention output, window-ordered rows, blocked layout, hi / lo planes
    int M;                                 // stream tokens (multiple of 16)
    T* xs; long long xs_plane;             // residual stream planes, blocked layout
    const int* winv;                       // stream token -> window row of the attention output
    const T *projh, *w1h, *w2h;            // fragment-order weights, hi plane only (prep_rowtile_weights / prep_mlp_weights with planes = 1)
    const float *proj_b, *g1, *e1, *b1, *b2, *g2, *e2;
    float eps;
};

// two consecutive 1 KiB fragments; the scheduling barrier pins the reads HERE in program order (fused_mlp.hip: ld_pair)
__device__ __forceinline__ void sk_ld2(const char* p, uint4 (&w)[2]) {
    w[0] = *reinterpret_cast<const uint4*>(p);
    w[1] = *reinterpret_cast<const uint4*>(p + 1024);
    __builtin_amdgcn_sched_barrier(0);
}

// NP fragment pairs at consecutive KiB of `st`, through a ring of RD register pairs read RD - 1 pairs ahead of their MFMAs
template <int NP, int RD, class Body>
__device__ __forceinline__ void sk_stream(const char* st, Body&& body) {
    uint4 ring[RD][2];
#pragma unroll
    for (int p = 0; p < RD - 1 && p < NP; ++p) sk_ld2(st + (p << 11), ring[p % RD]);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        if (p + RD - 1 < NP) sk_ld2(st + ((p + RD - 1) << 11), ring[(p + RD - 1) % RD]);
        body(p, ring[p % RD][0], ring[p % RD][1]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

template <class T, class S>
__global__ void __launch_bounds__(S::THREADS) __attribute__((amdgpu_waves_per_eu(2, 2)))
proj_mlp2_kernel(const Block2Args<T> a) {
    constexpr int C = S::C, FM = S::FM, KS = S::KS, CF = S::CF, NCH = S::NCH, NPB = S::NPB, NWAVES = S::NWAVES, RD = S::RD;
    constexpr bool P_GELU = S::PROBE & 1, P_LDS = S::PROBE & 2, P_DMA = S::PROBE & 4, P_BAR = S::PROBE & 8, P_IO = S::PROBE & 16;
    static_assert(CF / 2 == KS && (KS % 2) == 0, "fc1 and fc2 chunks hold the same, even number of fragment pairs (two-deep ring)");
    typedef typename OpT<T>::v8 v8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* tab = reinterpret_cast<float*>(smem + S::NSLOT * S::SLOT);
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds_base = (unsigned)(size_t)smem;
    const char* lrd = smem + lane * 16;                          // a lane's 16 bytes of every fragment

    // SLOT_KIB consecutive KiB of `src` -> LDS `dst`, piece q by wave q % 8 (wave-uniform base in SGPRs + the lane's 16-byte offset)
    const unsigned voff = lane * 16;
    auto dma1 = [&](const T* src, unsigned dst) {
#pragma unroll
        for (int i = 0; i < (S::SLOT_KIB + NWAVES - 1) / NWAVES; ++i) {
            const int q = wave + i * NWAVES;
            if (q < S::SLOT_KIB) glds16_s(src + (q << 9), voff, dst + (unsigned)(q << 10));
        }
    };
    // piece i (0 .. PPW-1) of this wave's share of the requests of interval j: W1(j + 1) -> slot (j + 1) & 1, W2(j) -> slot 2 + (j & 1)
    auto dma_piece = [&](int j, int i) {
        const int q = wave + i * NWAVES;
        const bool first = (S::SLOT_KIB % NWAVES == 0) ? (i < S::SLOT_KIB / NWAVES) : (q < S::SLOT_KIB);
        const int jj = first ? j + 1 : j, r = first ? q : q - S::SLOT_KIB;
        if (jj >= NCH) return;
        const T* src = (first ? a.w1h : a.w2h) + (((long long)jj * S::SLOT_KIB + r) << 9);
        glds16_s(src, voff, lds_base + (unsigned)(((first ? 0 : 2) + (jj & 1)) * S::SLOT + (r << 10)));
    };
    dma1(a.projh, lds_base + 2 * S::SLOT);
    dma1(a.w1h, lds_base);
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
        const int src = live[t] && !P_IO ? a.winv[(rb0 + t) * 16 + l15] : 0;
        const T* p = a.ao + blk_off(src, g * 8, C);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if constexpr (P_IO) { xh[t][ks] = v8{}; xl[t][ks] = v8{}; xh[t][ks][0] = (T)(float)lane; continue; }
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

    // ---- phase 1: projection, blocks of 32 output columns alternating between slots 2 and 3 ------------------------------------- //
#pragma unroll
    for (int j = 0; j < NPB; ++j) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                               // block j landed; every wave is done with block j - 1
        if (j + 1 < NPB) dma1(a.projh + ((long long)(j + 1) * S::SLOT_KIB << 9), lds_base + (unsigned)((2 + ((j + 1) & 1)) * S::SLOT));
        sk_stream<KS, RD>(lrd + (2 + (j & 1)) * S::SLOT, [&](int ks, const uint4& w0, const uint4& w1) {
#pragma unroll
            for (int t = 0; t < FM; ++t) yacc[t][2 * j] = OpT<T>::mfma(as_v8<T>(w0), xl[t][ks], yacc[t][2 * j]);
#pragma unroll
            for (int t = 0; t < FM; ++t) yacc[t][2 * j + 1] = OpT<T>::mfma(as_v8<T>(w1), xl[t][ks], yacc[t][2 * j + 1]);
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
            if constexpr (P_IO) { oh[bp] = xh[t][bp]; ol[bp] = xl[t][bp]; continue; }
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

    // ---- phase 2: the MLP, software-pipelined: interval j = [fc1(j) + GELU(j - 1)] [fc2(j - 1) + DMA requests], one barrier ---------- //
    f32x4 hacc[FM][2], hprev[FM][2];
    uint4 hh[FM], hl[FM];
    f32x2 vt[FM][4];
    float4 bb0, bb1;
    // bias + GELU + hi / lo split of the previous chunk in 18 FM small work items (5-10 VALU instructions each), so that every fc1 step
    // gets its share: per fragment t and value pair k four stages of gelu_erf2e (common.h: the Estrin form of the erfc fit), then the split
    // into fc2's operand registers in two halves (the lane's 8 hidden units 16 n + 4 g + r are k-slots 8 g + 4 n + r of fc2)
    constexpr int NITEMS = 18 * FM;
    f32x2 gx, ga, g0, g1, g2, g3, g4;                    // the pair in flight
    // An item's arithmetic depends on nothing the MFMAs of its step produce, and hipcc's instruction selection is free to emit it at
    // the top of the loop body (it does: all of the GELU in front of the first MFMA).  The empty volatile asm on an item's input keeps
    // its place among the scheduling barriers, and everything that depends on its output stays behind it.
    auto item = [&](int i) {
        const int t = i / 18, r = i % 18;
        if (r < 16) {
            const int k = r >> 2, st = r & 3;
            if (st == 0) {
                asm volatile("" : "+v"(hprev[t][k >> 1]));
                const f32x4& h = hprev[t][k >> 1];
                const float b0 = (k & 2) ? ((k & 1) ? bb1.z : bb1.x) : ((k & 1) ? bb0.z : bb0.x), b1 = (k & 2) ? ((k & 1) ? bb1.w : bb1.y) : ((k & 1) ? bb0.w : bb0.y);
                gx = f32x2{h[2 * (k & 1)] + b0, h[2 * (k & 1) + 1] + b1};
                if constexpr (P_GELU) return;
                ga.x = __builtin_amdgcn_fmed3f(__builtin_fabsf(gx.x), 0.f, 5.9396970f);
                ga.y = __builtin_amdgcn_fmed3f(__builtin_fabsf(gx.y), 0.f, 5.9396970f);
                g0 = ga * ga;
                g1 = __builtin_elementwise_fma(ga, (f32x2)(-4.591012597e-01f), (f32x2)(-1.151117682e+00f));
                g2 = __builtin_elementwise_fma(ga, (f32x2)(7.545167115e-03f), (f32x2)(-5.282834917e-02f));
            } else if (st == 1) {
                if constexpr (P_GELU) return;
                asm volatile("" : "+v"(ga));
                g3 = __builtin_elementwise_fma(ga, (f32x2)(-4.273382365e-05f), (f32x2)(-4.982745158e-04f));
                g4 = __builtin_elementwise_fma(ga, (f32x2)(-6.177511978e-07f), (f32x2)(1.091520153e-05f));
                g1 = __builtin_elementwise_fma(g0, g2, g1);
                g3 = __builtin_elementwise_fma(g0, g4, g3);
                g0 = g0 * g0;
            } else if (st == 2) {
                if constexpr (P_GELU) return;
                asm volatile("" : "+v"(g0));
                g1 = __builtin_elementwise_fma(g0, g3, g1) * ga;
                g2.x = __builtin_amdgcn_exp2f(g1.x);
                g2.y = __builtin_amdgcn_exp2f(g1.y);
            } else {
                if constexpr (P_GELU) { vt[t][k] = gx; return; }
                asm volatile("" : "+v"(g2));
                f32x2 e;
                e.x = __builtin_copysignf(1.0f - g2.x, gx.x);
                e.y = __builtin_copysignf(1.0f - g2.y, gx.y);
                vt[t][k] = ((f32x2)(0.5f) * gx) * ((f32x2)(1.0f) + e);
            }
            return;
        }
        asm volatile("" : "+v"(vt[t][r == 16 ? 0 : 3]));
        const float v[8] = {vt[t][0].x, vt[t][0].y, vt[t][1].x, vt[t][1].y, vt[t][2].x, vt[t][2].y, vt[t][3].x, vt[t][3].y};
        if (r == 16) {
            uint4 o[1];
            split8<T, 1>(v, o);
            hh[t] = o[0];
        } else {
            const v8 h = as_v8<T>(hh[t]);
            T l[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) l[e] = (T)(v[e] - (float)h[e]);
            hl[t] = __builtin_bit_cast(uint4, *reinterpret_cast<v8*>(l));
        }
    };
    auto fc1_step = [&](int ks, const uint4& w0, const uint4& w1) {
        if (ks == 0) {
#pragma unroll
            for (int t = 0; t < FM; ++t) { hacc[t][0] = f32x4{0.f, 0.f, 0.f, 0.f}; hacc[t][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        }
#pragma unroll
        for (int t = 0; t < FM; ++t) hacc[t][0] = OpT<T>::mfma(as_v8<T>(w0), xl[t][ks], hacc[t][0]);
#pragma unroll
        for (int t = 0; t < FM; ++t) hacc[t][1] = OpT<T>::mfma(as_v8<T>(w1), xl[t][ks], hacc[t][1]);
#pragma unroll
        for (int t = 0; t < FM; ++t) hacc[t][0] = OpT<T>::mfma(as_v8<T>(w0), xh[t][ks], hacc[t][0]);
#pragma unroll
        for (int t = 0; t < FM; ++t) hacc[t][1] = OpT<T>::mfma(as_v8<T>(w1), xh[t][ks], hacc[t][1]);
    };
    auto fc2_step = [&](int p, const uint4& w0, const uint4& w1) {
#pragma unroll
        for (int t = 0; t < FM; ++t) yacc[t][2 * p] = OpT<T>::mfma(as_v8<T>(w0), as_v8<T>(hl[t]), yacc[t][2 * p]);
#pragma unroll
        for (int t = 0; t < FM; ++t) yacc[t][2 * p + 1] = OpT<T>::mfma(as_v8<T>(w1), as_v8<T>(hl[t]), yacc[t][2 * p + 1]);
#pragma unroll
        for (int t = 0; t < FM; ++t) yacc[t][2 * p] = OpT<T>::mfma(as_v8<T>(w0), as_v8<T>(hh[t]), yacc[t][2 * p]);
#pragma unroll
        for (int t = 0; t < FM; ++t) yacc[t][2 * p + 1] = OpT<T>::mfma(as_v8<T>(w1), as_v8<T>(hh[t]), yacc[t][2 * p + 1]);
    };
    auto top = [&]() {
        if constexpr (!P_BAR) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    };
    auto load_bias = [&](int j) {
        bb0 = *reinterpret_cast<const float4*>(tab + S::T_B1 + j * 32 + 4 * g);
        bb1 = *reinterpret_cast<const float4*>(tab + S::T_B1 + j * 32 + 16 + 4 * g);
    };
    auto roll = [&]() {
#pragma unroll
        for (int t = 0; t < FM; ++t) { hprev[t][0] = hacc[t][0]; hprev[t][1] = hacc[t][1]; }
    };
    auto ld2 = [&](const char* p, uint4 (&w)[2]) {      // two consecutive fragments, NOT pinned: the group pipeline below places them
        w[0] = *reinterpret_cast<const uint4*>(p);
        w[1] = *reinterpret_cast<const uint4*>(p + 1024);
    };

    // interval 0: fc1(0) alone; W1(1) and W2(0) requested between its MFMAs
    top();                                              // W1(0) landed; every wave is done with the projection's last block
    sk_stream<KS, RD>(lrd, [&](int ks, const uint4& w0, const uint4& w1) {
        fc1_step(ks, w0, w1);
        if constexpr (!P_DMA) { if (ks < S::PPW) dma_piece(0, ks); }
    });
    roll();
    for (int j = 1; j < NCH; ++j) {
        top();                                          // W1(j), W2(j - 1) landed; every wave is done with interval j - 1
        load_bias(j - 1);
        __builtin_amdgcn_sched_barrier(0);
        const char* stA = lrd + (j & 1) * S::SLOT;
        const char* stB = lrd + (2 + ((j - 1) & 1)) * S::SLOT;
        uint4 ring[2][2];
        // phase A: fc1(j), fragment pairs read one pair ahead of their MFMAs (pinned), and behind the MFMAs of every step its share of the
        // GELU of chunk j - 1 (one MFMA, NV VALU, one MFMA, ...)
        sk_ld2(stA, ring[0]);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if constexpr (!P_LDS) { if (ks + 1 < KS) sk_ld2(stA + ((ks + 1) << 11), ring[(ks + 1) & 1]); else sk_ld2(stB, ring[(ks + 1) & 1]); }
            fc1_step(ks, ring[P_LDS ? 0 : ks & 1][0], ring[P_LDS ? 0 : ks & 1][1]);
#pragma unroll
            for (int i = 0; i < NITEMS; ++i)
                if ((i * KS) / NITEMS == ks) item(i);
            if constexpr (S::NV > 0) {
#pragma unroll
                for (int k = 0; k < 4 * FM; ++k) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, S::NV, 0); }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // phase B: fc2(j - 1) on the operand the GELU has just produced; the ring keeps running (its first pair was read under fc1's last
        // MFMAs); the LDS-DMA requests of W1(j + 1) and W2(j) ride behind the first steps
        if constexpr (P_LDS) { ld2(stB, ring[0]); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
        for (int p = 0; p < KS; ++p) {
            if constexpr (!P_LDS) { if (p + 1 < KS) sk_ld2(stB + ((p + 1) << 11), ring[(KS + p + 1) & 1]); }
            fc2_step(p, ring[P_LDS ? 0 : (KS + p) & 1][0], ring[P_LDS ? 0 : (KS + p) & 1][1]);
            if constexpr (!P_DMA) { if (p < S::PPW) dma_piece(j, p); }
            __builtin_amdgcn_sched_barrier(0);
        }
        roll();
    }
    top();                                              // W2(NCH - 1) landed
    load_bias(NCH - 1);
#pragma unroll
    for (int i = 0; i < NITEMS; ++i) item(i);
    sk_stream<KS, RD>(lrd + (2 + ((NCH - 1) & 1)) * S::SLOT, fc2_step);

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
        if (!live[t] || (P_IO && rstd != 12345.f)) continue;
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

// SKP_BLK2_VARIANT (measurement only): 0 default | 1 no forced splice (NV = 0) | 2 NV = 2 | 3 NV = 6 | 4 ring depth 3 | 10.. timing probes (C = 384)
hipError_t op_proj_mlp_skew(const Geom& g, const BlockW<f16>& b, const int* winv, int res, f16* Xs, const Work<PrecF16x3>& wk, hipStream_t s) {
    typedef f16 T;
    Block2Args<T> a{wk.ao, wk.ao_plane, g.ntok[res], Xs, wk.xs_plane[res], winv, b.projh, b.w1h, b.w2h,
                    b.proj_b, b.n1_g, b.n1_b, b.fc1_b, b.fc2_b, b.n2_g, b.n2_b, 1e-5f};
    if (a.M % 16 != 0) return hipErrorInvalidValue;
    static const int variant = [] { const char* v = getenv("SKP_BLK2_VARIANT"); return v ? atoi(v) : 0; }();
    if (res == 0) {
        switch (variant) {
            case 1: return launch_blk2<T, Blk2Shape<192, 2, 2, 0>>(a, s);
            case 2: return launch_blk2<T, Blk2Shape<192, 2, 2, 2>>(a, s);
            case 3: return launch_blk2<T, Blk2Shape<192, 2, 2, 6>>(a, s);
            case 4: return launch_blk2<T, Blk2Shape<192, 2, 3, 4>>(a, s);
            default: return launch_blk2<T, Blk2Shape<192, 2, 2, 4>>(a, s);
        }
    }
    switch (variant) {
        case 1: return launch_blk2<T, Blk2Shape<384, 1, 2, 0>>(a, s);
        case 2: return launch_blk2<T, Blk2Shape<384, 1, 2, 2>>(a, s);
        case 3: return launch_blk2<T, Blk2Shape<384, 1, 2, 6>>(a, s);
        case 4: return launch_blk2<T, Blk2Shape<384, 1, 3, 4>>(a, s);
        case 10: return launch_blk2<T, Blk2Shape<384, 1, 2, 4, 1>>(a, s);
        case 11: return launch_blk2<T, Blk2Shape<384, 1, 2, 4, 2>>(a, s);
        case 12: return launch_blk2<T, Blk2Shape<384, 1, 2, 4, 4>>(a, s);
        case 13: return launch_blk2<T, Blk2Shape<384, 1, 2, 4, 8>>(a, s);
        case 14: return launch_blk2<T, Blk2Shape<384, 1, 2, 4, 16>>(a, s);
        case 15: return launch_blk2<T, Blk2Shape<384, 1, 2, 4, 15>>(a, s);
        case 16: return launch_blk2<T, Blk2Shape<384, 1, 2, 4, 31>>(a, s);
        default: return launch_blk2<T, Blk2Shape<384, 1, 2, 4>>(a, s);
    }
}

}  // namespace skp
