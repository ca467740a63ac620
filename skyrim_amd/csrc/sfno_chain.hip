// Pixel-wise chains of the SFNO step as ONE kernel each (include/skyrim_sfno.h: sksfno_chain_run).
//
// Every 1x1 convolution of the network acts on one pixel's channel vector.  Run GEMM by GEMM (sfno_ops.hip) each of them reads
// and writes the whole [C][H][W] activation in fp32: at 721 x 1440 the encoder, the last block's MLP and the decoder alone move
// 16 GB per step.  Here a chain of them is one pass over the pixels:
//
//   ENC   x (raw state)            -> GELU(W1 n(x) + b1) -> W2 . + pos_embed                                    -> features
//   MLP   y (block, before norm1)  -> GELU(W1 norm1(y) + b1) -> W2 . + b2 + residual                            -> block output
//   TAIL  the last block's MLP, then the decoder on concat(block output, n(x)): GELU(V1 . + d1) -> V2 . + d2    -> next state
//
// (n = input normalisation, norm1 = instance norm: both are per-channel affines applied while the operand is loaded; the
// statistics of norm1 come from sksfno_instance_stats.)  The schedule is fused_mlp.hip's: a wavefront owns FM x 16 pixels for the
// whole kernel, their channel vectors live in registers as MFMA B-operand fragments (fp16 hi/lo pairs, 3 MFMA terms, fp32
// accumulate), the hidden layer of an expand/contract pair is walked in chunks of 32 units whose accumulators ARE, after bias +
// GELU + split, the operand fragments of the contracting layer, and LDS holds nothing but weights in fragment order, streamed by
// LDS-DMA.  With the perm8 row order of the contracting weights a lane's accumulators of fragment pair bp are channels
// 32 bp + 8 (lane >> 4) + [0..7]: exactly an input fragment of the NEXT pair, so TAIL chains two pairs without leaving registers.
// Activations are [C][HW] fp32 (pixels contiguous): a fragment is fetched with 8 dword loads per lane (16 consecutive pixels
// of 4 x 8 channels per instruction) and stored the same way.  HW must be a multiple of 16.  gfx950 only.
#include <cstdlib>

#include "gemm_dma.h"
#include "../../include/skyrim_sfno.h"

namespace skp {

constexpr int cmax(int a, int b) { return a > b ? a : b; }

template <int MODE_, int CP_, int HP_, int KXP_, int OP_, int FM_ = 1, int NWAVES_ = 8, int WPE_ = 0>
struct ChainShape {
    static constexpr int MODE = MODE_, CP = CP_, HP = HP_, KXP = KXP_, OP = OP_, FM = FM_, NWAVES = NWAVES_;
    static constexpr int WPE = WPE_ ? WPE_ : NWAVES / 4;     // waves per SIMD the kernel is compiled for (2 with 4 waves = two workgroups per CU)
    static constexpr int THREADS = 64 * NWAVES, BM = NWAVES * FM * 16;
    static constexpr bool TAIL = MODE == SKSFNO_CHAIN_TAIL;
    // pair 0:  K0 -> H0 -> CP        (ENC: the state's channels -> embed -> embed;  MLP / TAIL: embed -> hidden -> embed)
    static constexpr int K0 = MODE == SKSFNO_CHAIN_ENC ? KXP : CP, H0 = MODE == SKSFNO_CHAIN_ENC ? CP : HP;
    static constexpr int KS0 = K0 / 32, NCH0 = H0 / 32, CF0 = CP / 16;
    // pair 1 (TAIL): concat(embed, state) -> embed -> output channels
    static constexpr int KS1 = (CP + KXP) / 32, NCH1 = CP / 32, CF1 = OP / 16;
    // LDS stages (KiB): pair 0 keeps its W1 chunk at [0, A0) and its W2 chunk at [A0, A0 + B0); pair 1 reuses the same space as
    // [0, A1) and [A1, A1 + B1) -- its first W1 chunk is requested only after every wave has left pair 0
    static constexpr int A0 = KS0 * 4, B0 = CF0 * 2, A1 = KS1 * 4, B1 = CF1 * 2;
    static constexpr int LDS_BLK = cmax(A0 + B0, TAIL ? A1 + B1 : 0);
    // table (floats): scale[K0] shift[K0] b1[H0] b2[CP]  (+ TAIL: xscale[KXP] xshift[KXP] d1[CP] d2[OP])
    static constexpr int T_SCALE = 0, T_SHIFT = K0, T_B1 = 2 * K0, T_B2 = T_B1 + H0;
    static constexpr int T_XSCALE = T_B2 + CP, T_XSHIFT = T_XSCALE + KXP, T_D1 = T_XSHIFT + KXP, T_D2 = T_D1 + CP;
    static constexpr int TAB = TAIL ? T_D2 + OP : T_B2 + CP;
    static constexpr int SMEM = LDS_BLK * 1024 + TAB * 4;
    static_assert(CP % 32 == 0 && HP % 32 == 0 && KXP % 32 == 0 && OP % 32 == 0, "padded widths are multiples of 32");
    static_assert(SMEM <= 160 * 1024, "LDS");
};

struct ChainArgs {
    const float* y;       // pair 0 input, [K0 real][HW]
    const float* x;       // TAIL: raw state [KX][HW]
    const float* res;     // [C][HW]: residual (MLP / TAIL), position embedding (ENC)
    float* out;           // [C][HW] (ENC / MLP), [OUT][HW] (TAIL)
    long long HW;
    int C, KX, OUT;       // real channel counts
    const f16 *w1f, *w2f, *v1f, *v2f;
    const float* tab;
};

typedef OpT<f16>::v8 v8;

__device__ __forceinline__ void ch_ld_pair(const char* p, uint4 (&w)[2]) {
    w[0] = *reinterpret_cast<const uint4*>(p);
    w[1] = *reinterpret_cast<const uint4*>(p + 1024);
    __builtin_amdgcn_sched_barrier(0);       // keep the reads HERE, ahead of the MFMAs that follow (fused_mlp.hip)
}

// NBLK consecutive KiB blocks of a fragment-order weight array -> LDS, block b by wave b % NWAVES
template <int NBLK, int NWAVES>
__device__ __forceinline__ void dma_blocks(const f16* src, unsigned lds_dst, int wave, int lane) {
    const f16* s = src + lane * 8;
#pragma unroll
    for (int i = 0; i < (NBLK + NWAVES - 1) / NWAVES; ++i) {
        const int b = wave + i * NWAVES;
        if (NBLK % NWAVES == 0 || b < NBLK) glds16(s + (b << 9), lds_dst + (unsigned)(b << 10));
    }
}

// channels [0, 32 KS) of FM x 16 pixels as B-operand fragments KOFF .. KOFF + KS - 1: lane (l15, g) holds, for pixel l15 of
// fragment row t, channels 32 ks + 8 g + [0..7], after the per-channel affine, split into fp16 hi / lo
template <int FM, int KS, int KTOT, int KOFF>
__device__ __forceinline__ void load_frags(const float* src, long long HW, int creal, long long p0, const bool (&live)[FM],
                                           const float* scale, const float* shift, int lane, v8 (&xh)[FM][KTOT], v8 (&xl)[FM][KTOT]) {
    const int l15 = lane & 15, g = lane >> 4;
#pragma unroll
    for (int t = 0; t < FM; ++t) {
        const long long pix = live[t] ? p0 + 16 * t + l15 : 0;
        float v[KS][8];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int ch = 32 * ks + 8 * g + e;
                v[ks][e] = src[(long long)(ch < creal ? ch : 0) * HW + pix];
            }
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int c0 = 32 * ks + 8 * g;
            const float4 s0 = *reinterpret_cast<const float4*>(scale + c0), s1 = *reinterpret_cast<const float4*>(scale + c0 + 4);
            const float4 h0 = *reinterpret_cast<const float4*>(shift + c0), h1 = *reinterpret_cast<const float4*>(shift + c0 + 4);
            const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w}, sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
            float w[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) w[e] = (c0 + e < creal) ? v[ks][e] * sc[e] + sh[e] : 0.f;
            uint4 o[2];
            split8<f16, 2>(w, o);
            xh[t][KOFF + ks] = as_v8<f16>(o[0]);
            xl[t][KOFF + ks] = as_v8<f16>(o[1]);
        }
    }
}

// One expand / contract pair over the wave's pixels:  yacc += W2 GELU(W1 x + b1)   (W1: [32 NCH][32 KS], W2: [16 CF][32 NCH]).
// On entry chunk 0 of w1f is on its way into stage A; ``after_last`` runs once stage A is free for good (next pair's first block).
template <class S, int KS, int NCH, int CF, class After>
__device__ __forceinline__ void run_pair(const v8 (&xh)[S::FM][KS], const v8 (&xl)[S::FM][KS], f32x4 (&yacc)[S::FM][CF], const f16* w1f, const f16* w2f,
                                         const float* b1, const char* stA, const char* stB, unsigned ldsA, unsigned ldsB, int wave, int lane, After&& after_last) {
    constexpr int FM = S::FM, DEPTH = 3, NS = KS * 2, W1_BLK = KS * 4, W2_BLK = CF * 2;
    const int g = lane >> 4;
    for (int j = 0; j < NCH; ++j) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                               // W1 block j landed; every wave is done with W2 block j - 1
        dma_blocks<W2_BLK, S::NWAVES>(w2f + ((long long)j * W2_BLK << 9), ldsB, wave, lane);
        f32x4 hacc[FM][2];
#pragma unroll
        for (int t = 0; t < FM; ++t) { hacc[t][0] = f32x4{0.f, 0.f, 0.f, 0.f}; hacc[t][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        {
            uint4 ring[DEPTH][2];
#pragma unroll
            for (int s = 0; s < DEPTH - 1 && s < NS; ++s) ch_ld_pair(stA + ((s * 2) << 10) + lane * 16, ring[s % DEPTH]);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                if (s + DEPTH - 1 < NS) ch_ld_pair(stA + (((s + DEPTH - 1) * 2) << 10) + lane * 16, ring[(s + DEPTH - 1) % DEPTH]);
                const int ks = s >> 1, n = s & 1;
                const uint4 wh = ring[s % DEPTH][0], wl = ring[s % DEPTH][1];
#pragma unroll
                for (int t = 0; t < FM; ++t) hacc[t][n] = OpT<f16>::mfma(as_v8<f16>(wl), xh[t][ks], hacc[t][n]);
#pragma unroll
                for (int t = 0; t < FM; ++t) hacc[t][n] = OpT<f16>::mfma(as_v8<f16>(wh), xl[t][ks], hacc[t][n]);
#pragma unroll
                for (int t = 0; t < FM; ++t) hacc[t][n] = OpT<f16>::mfma(as_v8<f16>(wh), xh[t][ks], hacc[t][n]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // bias + GELU + hi/lo split: the lane's hidden units 16 n + 4 g + r of the chunk are k-slots 4 n + r of the contracting layer
        uint4 hh[FM], hl[FM];
        {
            const float4 bb0 = *reinterpret_cast<const float4*>(b1 + j * 32 + 4 * g), bb1 = *reinterpret_cast<const float4*>(b1 + j * 32 + 16 + 4 * g);
#pragma unroll
            for (int t = 0; t < FM; ++t) {
                const f32x2 a0 = gelu_erf2(f32x2{hacc[t][0][0] + bb0.x, hacc[t][0][1] + bb0.y}), a1 = gelu_erf2(f32x2{hacc[t][0][2] + bb0.z, hacc[t][0][3] + bb0.w});
                const f32x2 a2 = gelu_erf2(f32x2{hacc[t][1][0] + bb1.x, hacc[t][1][1] + bb1.y}), a3 = gelu_erf2(f32x2{hacc[t][1][2] + bb1.z, hacc[t][1][3] + bb1.w});
                const float v[8] = {a0.x, a0.y, a1.x, a1.y, a2.x, a2.y, a3.x, a3.y};
                uint4 o[2];
                split8<f16, 2>(v, o);
                hh[t] = o[0]; hl[t] = o[1];
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                               // W2 block j landed; every wave is done with W1 block j
        if (j + 1 < NCH) dma_blocks<W1_BLK, S::NWAVES>(w1f + ((long long)(j + 1) * W1_BLK << 9), ldsA, wave, lane);
        else after_last();
        {
            uint4 ring[DEPTH][2];
#pragma unroll
            for (int c = 0; c < DEPTH - 1 && c < CF; ++c) ch_ld_pair(stB + ((c * 2) << 10) + lane * 16, ring[c % DEPTH]);
#pragma unroll
            for (int c = 0; c < CF; ++c) {
                if (c + DEPTH - 1 < CF) ch_ld_pair(stB + (((c + DEPTH - 1) * 2) << 10) + lane * 16, ring[(c + DEPTH - 1) % DEPTH]);
                const uint4 wh = ring[c % DEPTH][0], wl = ring[c % DEPTH][1];
#pragma unroll
                for (int t = 0; t < FM; ++t) yacc[t][c] = OpT<f16>::mfma(as_v8<f16>(wl), as_v8<f16>(hh[t]), yacc[t][c]);
#pragma unroll
                for (int t = 0; t < FM; ++t) yacc[t][c] = OpT<f16>::mfma(as_v8<f16>(wh), as_v8<f16>(hl[t]), yacc[t][c]);
#pragma unroll
                for (int t = 0; t < FM; ++t) yacc[t][c] = OpT<f16>::mfma(as_v8<f16>(wh), as_v8<f16>(hh[t]), yacc[t][c]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}

template <class S>
__global__ void __launch_bounds__(S::THREADS) __attribute__((amdgpu_waves_per_eu(S::WPE, S::WPE)))
sfno_chain_kernel(const ChainArgs a) {
    constexpr int FM = S::FM, KS0 = S::KS0, CF0 = S::CF0;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* tab = reinterpret_cast<float*>(smem + S::LDS_BLK * 1024);
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = (unsigned)(size_t)smem;

    dma_blocks<KS0 * 4, S::NWAVES>(a.w1f, lds0, wave, lane);
    for (int i = tid; i < S::TAB; i += S::THREADS) tab[i] = a.tab[i];
    __syncthreads();

    const long long p0 = (long long)blockIdx.x * S::BM + (long long)wave * FM * 16;
    bool live[FM];
#pragma unroll
    for (int t = 0; t < FM; ++t) live[t] = p0 + 16 * t < a.HW;

    v8 xh[FM][KS0], xl[FM][KS0];
    load_frags<FM, KS0, KS0, 0>(a.y, a.HW, S::MODE == SKSFNO_CHAIN_ENC ? a.KX : a.C, p0, live, tab + S::T_SCALE, tab + S::T_SHIFT, lane, xh, xl);

    f32x4 yacc[FM][CF0];
#pragma unroll
    for (int t = 0; t < FM; ++t)
#pragma unroll
        for (int c = 0; c < CF0; ++c) yacc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    run_pair<S, KS0, S::NCH0, CF0>(xh, xl, yacc, a.w1f, a.w2f, tab + S::T_B1, smem, smem + S::A0 * 1024, lds0, lds0 + S::A0 * 1024, wave, lane, [] {});

    if constexpr (!S::TAIL) {
        // + b2 + residual (ENC: position embedding), store channel-major.  All loads before the first store.
#pragma unroll
        for (int t = 0; t < FM; ++t) {
            if (!live[t]) continue;
            const long long pix = p0 + 16 * t + l15;
            float r[S::CP / 32][8];
#pragma unroll
            for (int bp = 0; bp < S::CP / 32; ++bp)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int ch = 32 * bp + 8 * g + i;
                    r[bp][i] = a.res[(long long)(ch < a.C ? ch : 0) * a.HW + pix];
                }
#pragma unroll
            for (int bp = 0; bp < S::CP / 32; ++bp) {
                const int c0 = 32 * bp + 8 * g;
                const float4 b0 = *reinterpret_cast<const float4*>(tab + S::T_B2 + c0), b1 = *reinterpret_cast<const float4*>(tab + S::T_B2 + c0 + 4);
                const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float acc = i < 4 ? yacc[t][2 * bp][i] : yacc[t][2 * bp + 1][i - 4];
                    if (c0 + i < a.C) a.out[(long long)(c0 + i) * a.HW + pix] = acc + bb[i] + r[bp][i];
                }
            }
        }
    } else {
        __syncthreads();                               // every wave has left pair 0: its stages are free
        dma_blocks<S::KS1 * 4, S::NWAVES>(a.v1f, lds0, wave, lane);      // lands under the loads and the arithmetic below
        // block output = yacc + b2 + residual -> fragments 0 .. CP/32 - 1 of the decoder's input; the normalised state -> the rest
        constexpr int KS1 = S::KS1, CF1 = S::CF1;
        v8 zh[FM][KS1], zl[FM][KS1];
#pragma unroll
        for (int t = 0; t < FM; ++t) {
            const long long pix = live[t] ? p0 + 16 * t + l15 : 0;
            float r[S::CP / 32][8];
#pragma unroll
            for (int bp = 0; bp < S::CP / 32; ++bp)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int ch = 32 * bp + 8 * g + i;
                    r[bp][i] = a.res[(long long)(ch < a.C ? ch : 0) * a.HW + pix];
                }
#pragma unroll
            for (int bp = 0; bp < S::CP / 32; ++bp) {
                const int c0 = 32 * bp + 8 * g;
                const float4 b0 = *reinterpret_cast<const float4*>(tab + S::T_B2 + c0), b1 = *reinterpret_cast<const float4*>(tab + S::T_B2 + c0 + 4);
                const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float acc = i < 4 ? yacc[t][2 * bp][i] : yacc[t][2 * bp + 1][i - 4];
                    v[i] = (c0 + i < a.C) ? acc + bb[i] + r[bp][i] : 0.f;
                }
                uint4 o[2];
                split8<f16, 2>(v, o);
                zh[t][bp] = as_v8<f16>(o[0]);
                zl[t][bp] = as_v8<f16>(o[1]);
            }
        }
        load_frags<FM, S::KXP / 32, KS1, S::CP / 32>(a.x, a.HW, a.KX, p0, live, tab + S::T_XSCALE, tab + S::T_XSHIFT, lane, zh, zl);

        f32x4 zacc[FM][CF1];
#pragma unroll
        for (int t = 0; t < FM; ++t)
#pragma unroll
            for (int c = 0; c < CF1; ++c) zacc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        run_pair<S, KS1, S::NCH1, CF1>(zh, zl, zacc, a.v1f, a.v2f, tab + S::T_D1, smem, smem + S::A1 * 1024, lds0, lds0 + S::A1 * 1024, wave, lane, [] {});
#pragma unroll
        for (int t = 0; t < FM; ++t) {
            if (!live[t]) continue;
            const long long pix = p0 + 16 * t + l15;
#pragma unroll
            for (int bp = 0; bp < S::OP / 32; ++bp) {
                const int c0 = 32 * bp + 8 * g;
                const float4 b0 = *reinterpret_cast<const float4*>(tab + S::T_D2 + c0), b1 = *reinterpret_cast<const float4*>(tab + S::T_D2 + c0 + 4);
                const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float acc = i < 4 ? zacc[t][2 * bp][i] : zacc[t][2 * bp + 1][i - 4];
                    if (c0 + i < a.OUT) a.out[(long long)(c0 + i) * a.HW + pix] = acc + bb[i];
                }
            }
        }
    }
}

// ---- prepare: padded fp32 matrices -> fragment-order hi/lo planes ------------------------------------------------------------- //
//   w1f[((j KS + ks) 2 + n) 2 + plane][lane][e] = W1[32 j + 16 n + (lane & 15)][32 ks + 8 (lane >> 4) + e]          W1: [H][K]
//   w2f[(j CF + c) 2 + plane][lane][e]          = W2[perm8_col(16 c + (lane & 15))][32 j + 16 (e >> 2) + 4 (lane >> 4) + (e & 3)]   W2: [N][H]
__global__ void prep_chain_w1_kernel(const float* __restrict__ w1, f16* __restrict__ out, int H, int K) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int KS = K / 32;
    const long long total = (long long)(H / 32) * KS * 2 * 512;
    if (i >= total) return;
    const int e = (int)(i & 7), lane = (int)((i >> 3) & 63);
    long long q = i >> 9;
    const int n = (int)(q & 1); q >>= 1;
    const int ks = (int)(q % KS);
    const int j = (int)(q / KS);
    const float v = w1[(long long)(32 * j + 16 * n + (lane & 15)) * K + 32 * ks + 8 * (lane >> 4) + e];
    const f16 h = (f16)v;
    const long long o = ((((long long)j * KS + ks) * 2 + n) * 2 << 9) + lane * 8 + e;
    out[o] = h;
    out[o + 512] = (f16)(v - (float)h);
}

__global__ void prep_chain_w2_kernel(const float* __restrict__ w2, f16* __restrict__ out, int N, int H) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int CF = N / 16;
    const long long total = (long long)(H / 32) * CF * 512;
    if (i >= total) return;
    const int e = (int)(i & 7), lane = (int)((i >> 3) & 63);
    const long long q = i >> 9;
    const int c = (int)(q % CF);
    const int j = (int)(q / CF);
    const int row = perm8_col(16 * c + (lane & 15));
    const int hid = 32 * j + 16 * (e >> 2) + 4 * (lane >> 4) + (e & 3);
    const float v = w2[(long long)row * H + hid];
    const f16 h = (f16)v;
    const long long o = ((((long long)j * CF + c) * 2) << 9) + lane * 8 + e;
    out[o] = h;
    out[o + 512] = (f16)(v - (float)h);
}

// ---- instance-norm statistics as the per-channel affine of the consumer:  scale = gamma rstd,  shift = beta - mean gamma rstd ---- //
__global__ void __launch_bounds__(1024) instance_stats_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              float* __restrict__ scale, float* __restrict__ shift, long long HW, float eps) {
    __shared__ float red[2][16];
    const float* xc = x + (long long)blockIdx.x * HW;
    const float pv = xc[0];                          // shifted one-pass moments (sfno_ops.hip: instance_norm_kernel)
    float s = 0.f, q = 0.f;
    if ((HW & 3) == 0) {
        for (long long i = threadIdx.x; i < HW / 4; i += blockDim.x) {
            const float4 v = reinterpret_cast<const float4*>(xc)[i];
            const float a = v.x - pv, b = v.y - pv, c = v.z - pv, d = v.w - pv;
            s += (a + b) + (c + d);
            q += (a * a + b * b) + (c * c + d * d);
        }
    } else {
        for (long long i = threadIdx.x; i < HW; i += blockDim.x) { const float d = xc[i] - pv; s += d; q += d * d; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); q += __shfl_xor(q, o); }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[0][wave] = s; red[1][wave] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float ts = 0.f, tq = 0.f;
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) { ts += red[0][w]; tq += red[1][w]; }
        const float m1 = ts / (float)HW, m2 = tq / (float)HW;
        const float rstd = rsqrtf(fmaxf(m2 - m1 * m1, 0.f) + eps);
        const float gsc = gamma[blockIdx.x] * rstd;
        scale[blockIdx.x] = gsc;
        shift[blockIdx.x] = beta[blockIdx.x] - (pv + m1) * gsc;
    }
}

template <class S>
hipError_t launch_chain(const ChainArgs& a, hipStream_t s) {
    auto kern = sfno_chain_kernel<S>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, S::SMEM);
    if (e != hipSuccess) return e;
    const unsigned grid = (unsigned)((a.HW + S::BM - 1) / S::BM);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(S::THREADS), S::SMEM, s, a);
    return hipGetLastError();
}

// padded widths per shape class: {CP, HP, KXP, OP}
constexpr int kShapes[2][4] = {{256, 512, 96, 96}, {64, 96, 32, 32}};

// One 8-wave workgroup of 128 pixels per CU by default.  SKSFNO_CHAIN_WAVES=4: two independent 4-wave workgroups of 64 pixels per CU
// (twice the weight traffic per pixel) -- measured equal at 721 x 1440 (18.54 vs 18.56 ms/step): the chunk loop, not the operand
// loads / stores around it, sets the pace (with one chunk instead of all of them the three chains take 1.9 of their 4.9 ms).
template <int MODE>
hipError_t launch_mode(int shape, const ChainArgs& a, hipStream_t s) {
    static const int waves = [] { const char* v = getenv("SKSFNO_CHAIN_WAVES"); return v ? atoi(v) : 8; }();
    if (shape == 0) {
        if (waves == 8) return launch_chain<ChainShape<MODE, kShapes[0][0], kShapes[0][1], kShapes[0][2], kShapes[0][3], 1, 8>>(a, s);
        return launch_chain<ChainShape<MODE, kShapes[0][0], kShapes[0][1], kShapes[0][2], kShapes[0][3], 1, 4, 2>>(a, s);
    }
    if (waves == 8) return launch_chain<ChainShape<MODE, kShapes[1][0], kShapes[1][1], kShapes[1][2], kShapes[1][3], 1, 8>>(a, s);
    return launch_chain<ChainShape<MODE, kShapes[1][0], kShapes[1][1], kShapes[1][2], kShapes[1][3], 1, 4, 2>>(a, s);
}

}  // namespace skp

using namespace skp;

extern "C" {

int sksfno_chain_dims(int shape, int* cp, int* hp, int* kxp, int* op) {
    if (shape < 0 || shape > 1 || !cp || !hp || !kxp || !op) return SKSFNO_E_ARG;
    *cp = kShapes[shape][0]; *hp = kShapes[shape][1]; *kxp = kShapes[shape][2]; *op = kShapes[shape][3];
    return 0;
}

int sksfno_prepare_chain_weights(const float* w1, const float* w2, int K, int H, int N, void* w1f, void* w2f, void* stream) {
    if (!w1 || !w2 || !w1f || !w2f || K <= 0 || H <= 0 || N <= 0 || (K & 31) || (H & 31) || (N & 31)) return SKSFNO_E_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long long t1 = (long long)(H / 32) * (K / 32) * 2 * 512, t2 = (long long)(H / 32) * (N / 16) * 512;
    hipLaunchKernelGGL(prep_chain_w1_kernel, dim3((unsigned)((t1 + 255) / 256)), dim3(256), 0, s, w1, static_cast<f16*>(w1f), H, K);
    hipLaunchKernelGGL(prep_chain_w2_kernel, dim3((unsigned)((t2 + 255) / 256)), dim3(256), 0, s, w2, static_cast<f16*>(w2f), N, H);
    return hipGetLastError() == hipSuccess ? 0 : SKSFNO_E_HIP;
}

int sksfno_instance_stats(const float* x, const float* gamma, const float* beta, float* scale, float* shift, int C, long long HW, float eps, void* stream) {
    if (!x || !gamma || !beta || !scale || !shift || C <= 0 || HW <= 0) return SKSFNO_E_ARG;
    hipLaunchKernelGGL(instance_stats_kernel, dim3(C), dim3(1024), 0, static_cast<hipStream_t>(stream), x, gamma, beta, scale, shift, HW, eps);
    return hipGetLastError() == hipSuccess ? 0 : SKSFNO_E_HIP;
}

int sksfno_chain_run(const sksfno_chain* d, void* stream) {
    if (!d || d->shape < 0 || d->shape > 1 || !d->y || !d->res || !d->out || !d->w1f || !d->w2f || !d->tab || d->HW <= 0 || (d->HW & 15)) return SKSFNO_E_ARG;
    const int* sh = kShapes[d->shape];
    if (d->C <= 0 || d->C > sh[0] || d->KX < 0 || d->KX > sh[2] || d->OUT < 0 || d->OUT > sh[3]) return SKSFNO_E_ARG;
    ChainArgs a{d->y, d->x, d->res, d->out, d->HW, d->C, d->KX, d->OUT, static_cast<const f16*>(d->w1f), static_cast<const f16*>(d->w2f),
                static_cast<const f16*>(d->v1f), static_cast<const f16*>(d->v2f), d->tab};
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e;
    switch (d->mode) {
        case SKSFNO_CHAIN_ENC:
            if (d->KX <= 0) return SKSFNO_E_ARG;
            e = launch_mode<SKSFNO_CHAIN_ENC>(d->shape, a, s);
            break;
        case SKSFNO_CHAIN_MLP: e = launch_mode<SKSFNO_CHAIN_MLP>(d->shape, a, s); break;
        case SKSFNO_CHAIN_TAIL:
            if (!d->x || !d->v1f || !d->v2f || d->KX <= 0 || d->OUT <= 0) return SKSFNO_E_ARG;
            e = launch_mode<SKSFNO_CHAIN_TAIL>(d->shape, a, s);
            break;
        default: return SKSFNO_E_ARG;
    }
    return e == hipSuccess ? 0 : SKSFNO_E_HIP;
}

}  // extern "C"
