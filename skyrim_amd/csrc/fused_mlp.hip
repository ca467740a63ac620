// MLP half of an EarthSpecificBlock as ONE kernel:   x += LayerNorm(norm2)( fc2( GELU( fc1(x) ) ) )   in place on the residual planes.
//
// The two-kernel form (ops_mlp.hip) writes the 4C-wide hidden activation to HBM as hi/lo planes and reads it back: 32 of the
// step's 90 GB.  Here the hidden never leaves the CU, and neither do the token rows:
//
//   * a wavefront owns FM x 16 tokens for the whole kernel.  Their C input columns live in REGISTERS as MFMA B-operand fragments
//     (hi + lo planes, loaded once with 16-byte loads: one wave instruction = one 1 KiB block of the blocked layout), and so do
//     the FM x 16 x C fp32 accumulators of fc2 -- TM x C = 12288 either way (C = 192: 64 tokens, C = 384: 32 tokens), a
//     512-register wave, one wave per SIMD, 4 waves per workgroup.  Nothing of the token tile ever goes through LDS.
//   * the hidden dimension is walked in chunks of 32 units.  fc1 of a chunk leaves, in swapped order (D^T = W X^T), for token
//     l & 15 the hidden units 16 n + 4 (l >> 4) + r in a lane's accumulators -- after bias + GELU + hi/lo split those eight
//     values ARE the lane's B-operand fragment of fc2's 32-deep k-step (the prepared fc2 weights carry the matching column
//     order).  No shuffle, no LDS round trip: the same trick as P -> P V in attention.hip.
//   * LDS holds only weights: per chunk the 32 x C block of fc1 and the C x 32 block of fc2, each as hi/lo planes in FRAGMENT
//     order (prepared once: every 1 KiB = what one ds_read_b128 wave instruction consumes, lane-linear, so LDS-DMA lands it
//     conflict-free with no swizzle).  Two stages: fc2's block streams in under fc1's MFMAs, the next chunk's fc1 block under
//     fc2's.  Per chunk and CU: 49 / 98 KB of DMA against 4608 MFMA clocks (25 % / 49 % of the 43 B/clk DMA rate; the tiled
//     GEMMs sit at 58 %), 384 ds_read_b128 per wave (a third of the LDS read rate).
//   * epilogue: a token's C outputs sit in one lane quad's registers across the 4 lane groups -> LayerNorm needs two shfl_xor,
//     no LDS, no barrier.  With the perm8 row order of the fc2 weights the accumulators of fragment pair bp are columns
//     32 bp + 8 (l >> 4) + [0..7] -- exactly the columns of the input fragment of k-step bp, still in registers: the residual
//     add needs no load at all.  HBM traffic of the whole MLP: 4 B/element read + 4 B/element written + weights (L2).
//
// gfx950 only.  One workgroup per CU; grid = ceil(tokens / BM).
#include <cstdlib>
#include "gemm_dma.h"
#include "launchers.h"

namespace skp {

template <int C_, int FM_, int NWAVES_, int DEPTH_ = 3, bool PIPE_ = false, int VALU_PER_MFMA_ = 4, int WPE_ = 0, bool STAGGER_ = false>
struct MlpShape {
    static constexpr int WPE = WPE_ ? WPE_ : NWAVES_ / 4;   // waves per SIMD the kernel is compiled for (2 with 4 waves = two workgroups per CU)
    static constexpr bool STAGGER = STAGGER_;               // odd waves issue their LDS-DMA in the middle of a phase instead of at its start
    static constexpr int C = C_, FM = FM_, NWAVES = NWAVES_, THREADS = 64 * NWAVES_, DEPTH = DEPTH_, VALU_PER_MFMA = VALU_PER_MFMA_;
    static constexpr bool PIPE = PIPE_;
    static constexpr int KS = C / 32;             // 32-deep k-steps of fc1 = fragment pairs of the output
    static constexpr int CF = C / 16;             // 16-wide output fragments of fc2
    static constexpr int HID = 4 * C, NCH = HID / 32;
    static constexpr int BM = NWAVES * FM * 16;
    static constexpr int W1_BLK = KS * 2 * 2;     // 1 KiB blocks of a chunk's fc1 weights: [ks][n][plane]
    static constexpr int W2_BLK = CF * 2;         // ... fc2 weights: [c][plane]
    static constexpr int STAGE_A = W1_BLK * 1024, STAGE_B = W2_BLK * 1024;
    static constexpr int TAB_FLOATS = HID + 3 * C;   // fc1 bias | fc2 bias | gamma | beta
    static constexpr int SMEM = STAGE_A + STAGE_B + TAB_FLOATS * 4;
    static_assert(W1_BLK % NWAVES == 0 && W2_BLK % NWAVES == 0, "DMA blocks per wave");
    static_assert(SMEM <= 160 * 1024, "LDS");
};

template <class T>
struct MlpArgs {
    T* xs;                  // residual stream: hi plane, blocked layout [tokens/16][C/32][16][32]; lo plane at + plane
    long long plane;
    int M;                  // tokens (multiple of 16)
    const T* w1f;           // fc1 weights in fragment order (prep_mlp_weights)
    const T* w2f;
    const float *b1, *b2, *gamma, *beta;
    float eps;
};

// one hi/lo fragment pair (two consecutive KiB blocks).  The scheduling barrier pins the reads HERE in program order, ahead of the
// MFMAs that follow in the source: left alone, hipcc's scheduler sinks them back next to their first use (one pair reloaded in
// place: read, wait, MFMAs, read, ...), and with one wave per SIMD nothing else hides the LDS latency.
__device__ __forceinline__ void ld_pair(const char* p, uint4 (&w)[2]) {
    w[0] = *reinterpret_cast<const uint4*>(p);
    w[1] = *reinterpret_cast<const uint4*>(p + 1024);
    __builtin_amdgcn_sched_barrier(0);
}

template <class T, class S>
__global__ void __launch_bounds__(S::THREADS) __attribute__((amdgpu_waves_per_eu(S::WPE, S::WPE)))
fused_mlp_kernel(const MlpArgs<T> a) {
    constexpr int C = S::C, FM = S::FM, KS = S::KS, CF = S::CF, HID = S::HID, NCH = S::NCH, NWAVES = S::NWAVES;
    constexpr int DEPTH = S::DEPTH;   // weight-fragment pairs in flight per wave (register ring)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* stA = smem;
    char* stB = smem + S::STAGE_A;
    float* tab = reinterpret_cast<float*>(smem + S::STAGE_A + S::STAGE_B);
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds_base = (unsigned)(size_t)smem;

    // weight stream: chunk j = W1_BLK contiguous KiB of w1f, W2_BLK of w2f; wave w fetches blocks w, w + NWAVES, ...
    auto issue_w1 = [&](int j) {
        const T* src = a.w1f + ((long long)j * S::W1_BLK << 9) + lane * 8;
#pragma unroll
        for (int i = 0; i < S::W1_BLK / NWAVES; ++i) {
            const int b = wave + i * NWAVES;
            glds16(src + (b << 9), lds_base + (unsigned)(b << 10));
        }
    };
    auto issue_w2 = [&](int j) {
        const T* src = a.w2f + ((long long)j * S::W2_BLK << 9) + lane * 8;
#pragma unroll
        for (int i = 0; i < S::W2_BLK / NWAVES; ++i) {
            const int b = wave + i * NWAVES;
            glds16(src + (b << 9), lds_base + (unsigned)(S::STAGE_A + (b << 10)));
        }
    };
    issue_w1(0);

    for (int i = tid; i < HID; i += S::THREADS) tab[i] = a.b1[i];
    for (int i = tid; i < C; i += S::THREADS) { tab[HID + i] = a.b2[i]; tab[HID + C + i] = a.gamma[i]; tab[HID + 2 * C + i] = a.beta[i]; }

    // the wave's token rows as B-operand fragments: fragment (t, ks) = block (row block, ks) of the blocked layout
    const long long rb0 = (long long)blockIdx.x * (S::BM / 16) + wave * FM;
    typedef typename OpT<T>::v8 v8;
    v8 xh[FM][KS], xl[FM][KS];
    bool live[FM];
#pragma unroll
    for (int t = 0; t < FM; ++t) {
        live[t] = (rb0 + t) * 16 < a.M;
        const T* p = a.xs + ((live[t] ? rb0 + t : 0) * KS << 9) + l15 * 32 + g * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            xh[t][ks] = *reinterpret_cast<const v8*>(p + (ks << 9));
            xl[t][ks] = *reinterpret_cast<const v8*>(p + (ks << 9) + a.plane);
        }
    }

    // the fragments are consumed here once, so that hipcc's vmcnt waits for these loads sit BEFORE the loop: inside it they would
    // also wait for the LDS-DMA of the block in flight, which the compiler's counter bookkeeping does not know about
#pragma unroll
    for (int t = 0; t < FM; ++t)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) { asm volatile("" : "+v"(xh[t][ks])); asm volatile("" : "+v"(xl[t][ks])); }

    f32x4 yacc[FM][CF];
#pragma unroll
    for (int t = 0; t < FM; ++t)
#pragma unroll
        for (int c = 0; c < CF; ++c) yacc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};

    f32x4 hacc[FM][2];
    constexpr int NS = KS * 2;                           // fc1 step s = (ks, n): one hi/lo fragment pair, 3 FM MFMAs

    // fc1 of one chunk out of stage A into hacc.  Weight fragments come through a ring of DEPTH register pairs, loaded DEPTH - 1
    // steps ahead of their MFMAs (hipcc on its own reloads one pair in place: read, wait, MFMAs, read, ... -- and with one wave
    // per SIMD nothing else hides the LDS latency).  ``side(s)`` = VALU work of ANOTHER chunk spliced into step s: the MFMA pipe
    // is busy 16 clocks per instruction, a wave issues in order, so VALU placed between two MFMAs runs for free.
    auto fc1 = [&](auto&& side, auto&& mid) {
#pragma unroll
        for (int t = 0; t < FM; ++t) { hacc[t][0] = f32x4{0.f, 0.f, 0.f, 0.f}; hacc[t][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        uint4 ring[DEPTH][2];
#pragma unroll
        for (int s = 0; s < DEPTH - 1 && s < NS; ++s) ld_pair(stA + ((s * 2) << 10) + lane * 16, ring[s % DEPTH]);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (s + DEPTH - 1 < NS) ld_pair(stA + (((s + DEPTH - 1) * 2) << 10) + lane * 16, ring[(s + DEPTH - 1) % DEPTH]);
            const int ks = s >> 1, n = s & 1;
            const uint4 wh = ring[s % DEPTH][0], wl = ring[s % DEPTH][1];
            const bool busy = side(s);
            if (s == NS / 2) mid();
#pragma unroll
            for (int t = 0; t < FM; ++t) hacc[t][n] = OpT<T>::mfma(as_v8<T>(wl), xh[t][ks], hacc[t][n]);
#pragma unroll
            for (int t = 0; t < FM; ++t) hacc[t][n] = OpT<T>::mfma(as_v8<T>(wh), xl[t][ks], hacc[t][n]);
#pragma unroll
            for (int t = 0; t < FM; ++t) hacc[t][n] = OpT<T>::mfma(as_v8<T>(wh), xh[t][ks], hacc[t][n]);
            if (busy) {                                  // alternate: one MFMA, a few VALU, one MFMA, ...
#pragma unroll
                for (int k = 0; k < 3 * FM; ++k) { __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); __builtin_amdgcn_sched_group_barrier(0x002, S::VALU_PER_MFMA, 0); }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    uint4 hh[FM], hl[FM];
    auto fc2 = [&](auto&& mid) {
        uint4 ring[DEPTH][2];
#pragma unroll
        for (int c = 0; c < DEPTH - 1 && c < CF; ++c) ld_pair(stB + ((c * 2) << 10) + lane * 16, ring[c % DEPTH]);
#pragma unroll
        for (int c = 0; c < CF; ++c) {
            if (c + DEPTH - 1 < CF) ld_pair(stB + (((c + DEPTH - 1) * 2) << 10) + lane * 16, ring[(c + DEPTH - 1) % DEPTH]);
            const uint4 wh = ring[c % DEPTH][0], wl = ring[c % DEPTH][1];
            if (c == CF / 2) mid();
#pragma unroll
            for (int t = 0; t < FM; ++t) yacc[t][c] = OpT<T>::mfma(as_v8<T>(wl), as_v8<T>(hh[t]), yacc[t][c]);
#pragma unroll
            for (int t = 0; t < FM; ++t) yacc[t][c] = OpT<T>::mfma(as_v8<T>(wh), as_v8<T>(hl[t]), yacc[t][c]);
#pragma unroll
            for (int t = 0; t < FM; ++t) yacc[t][c] = OpT<T>::mfma(as_v8<T>(wh), as_v8<T>(hh[t]), yacc[t][c]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // bias + GELU + hi/lo split of a chunk: the lane's 8 hidden units 16 n + 4 g + r become k-slots 8 g + 4 n + r of fc2.
    // Work items per token fragment: 4 x (GELU of two values) + 1 x (split into the hi / lo operand registers).
    f32x2 vt[FM][4];
    auto take = [&](int j) {                             // hacc + fc1 bias -> vt (hacc is free for the next chunk afterwards)
        const float4 bb0 = *reinterpret_cast<const float4*>(tab + j * 32 + 4 * g), bb1 = *reinterpret_cast<const float4*>(tab + j * 32 + 16 + 4 * g);
#pragma unroll
        for (int t = 0; t < FM; ++t) {
            vt[t][0] = f32x2{hacc[t][0][0] + bb0.x, hacc[t][0][1] + bb0.y};
            vt[t][1] = f32x2{hacc[t][0][2] + bb0.z, hacc[t][0][3] + bb0.w};
            vt[t][2] = f32x2{hacc[t][1][0] + bb1.x, hacc[t][1][1] + bb1.y};
            vt[t][3] = f32x2{hacc[t][1][2] + bb1.z, hacc[t][1][3] + bb1.w};
        }
    };
    constexpr int NITEMS = FM * 5;
    auto item = [&](int i) {
        const int t = i / 5, k = i % 5;
        if (k < 4) { vt[t][k] = gelu_erf2(vt[t][k]); return; }
        const float v[8] = {vt[t][0].x, vt[t][0].y, vt[t][1].x, vt[t][1].y, vt[t][2].x, vt[t][2].y, vt[t][3].x, vt[t][3].y};
        uint4 o[2];
        split8<T, 2>(v, o);
        hh[t] = o[0]; hl[t] = o[1];
    };

    const bool late = S::STAGGER && (wave & 1);          // wave-uniform
    auto nop = [] {};
    if constexpr (!S::PIPE) {
        // plain schedule per chunk:  fc1(j) | GELU(j) | fc2(j), two barriers
        for (int j = 0; j < NCH; ++j) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                           // fc1 block j landed; every wave is done with fc2 block j - 1
            if (!late) issue_w2(j);
            fc1([](int) { return false; }, [&] { if (late) issue_w2(j); });
            take(j);
#pragma unroll
            for (int i = 0; i < NITEMS; ++i) item(i);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                           // fc2 block j landed; every wave is done with fc1 block j
            if (!late && j + 1 < NCH) issue_w1(j + 1);
            fc2([&] { if (late && j + 1 < NCH) issue_w1(j + 1); });
        }
    } else {
        // skewed schedule: fc1 runs one chunk ahead, so that the GELU of chunk j is spliced between the MFMAs of fc1(j + 1):
        //   fc1(0) | [fc1(1) + GELU(0)] | fc2(0) | [fc1(2) + GELU(1)] | fc2(1) | ...        (stage A: fc1 blocks, stage B: fc2 blocks)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        fc1([](int) { return false; }, nop);
        __syncthreads();                               // every wave is done with fc1 block 0
        issue_w1(1);
        for (int j = 0; j < NCH; ++j) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                           // fc1 block j + 1 landed; every wave is done with fc2 block j - 1
            issue_w2(j);
            take(j);
            if (j + 1 < NCH) {
                fc1([&](int s) {
                    bool any = false;
#pragma unroll
                    for (int i = 0; i < NITEMS; ++i)
                        if ((i * NS) / NITEMS == s) { item(i); any = true; }
                    return any;
                }, nop);
            } else {
#pragma unroll
                for (int i = 0; i < NITEMS; ++i) item(i);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                           // fc2 block j landed; every wave is done with fc1 block j + 1
            if (j + 2 < NCH) issue_w1(j + 2);
            fc2(nop);
        }
    }

    // epilogue: + fc2 bias, LayerNorm over the token's C columns (in-lane sums + two shuffles), + the input still in registers
    const float* tb2 = tab + HID;
    const float* tg = tab + HID + C;
    const float* tbe = tab + HID + 2 * C;
#pragma unroll
    for (int t = 0; t < FM; ++t) {
        float s = 0.f;
#pragma unroll
        for (int bp = 0; bp < KS; ++bp) {
            const int n = 32 * bp + 8 * g;
            const float4 b0 = *reinterpret_cast<const float4*>(tb2 + n), b1 = *reinterpret_cast<const float4*>(tb2 + n + 4);
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
            const float4 g0 = *reinterpret_cast<const float4*>(tg + n), g1 = *reinterpret_cast<const float4*>(tg + n + 4);
            const float4 e0 = *reinterpret_cast<const float4*>(tbe + n), e1 = *reinterpret_cast<const float4*>(tbe + n + 4);
            float oh[8], ol[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { oh[i] = (float)xh[t][bp][i]; ol[i] = (float)xl[t][bp][i]; }
            const f32x4 &x = yacc[t][2 * bp], &z = yacc[t][2 * bp + 1];
            const float v[8] = {(oh[0] + ol[0]) + ((x[0] - mean) * rstd * g0.x + e0.x), (oh[1] + ol[1]) + ((x[1] - mean) * rstd * g0.y + e0.y),
                                (oh[2] + ol[2]) + ((x[2] - mean) * rstd * g0.z + e0.z), (oh[3] + ol[3]) + ((x[3] - mean) * rstd * g0.w + e0.w),
                                (oh[4] + ol[4]) + ((z[0] - mean) * rstd * g1.x + e1.x), (oh[5] + ol[5]) + ((z[1] - mean) * rstd * g1.y + e1.y),
                                (oh[6] + ol[6]) + ((z[2] - mean) * rstd * g1.z + e1.z), (oh[7] + ol[7]) + ((z[3] - mean) * rstd * g1.w + e1.w)};
            store8_planes<T, 2>(dst + (bp << 9), a.plane, v);
        }
    }
}

// ---- prepare: fp32 master weights -> fragment-order hi/lo planes ---------------------------------------------------------------- //
//   (planes = 1: the hi plane only, blocks [(j KS + ks) 2 + n] and [j CF + c] -- fused_block2.hip)
//   w1f[((j KS + ks) 2 + n) 2 + plane][lane][e] = fc1.weight[32 j + 16 n + (lane & 15)][32 ks + 8 (lane >> 4) + e]
//   w2f[(j CF + c) 2 + plane][lane][e]          = fc2.weight[perm8_col(16 c + (lane & 15))][32 j + 16 (e >> 2) + 4 (lane >> 4) + (e & 3)]
template <class T>
__global__ void prep_mlp_w1_kernel(const float* __restrict__ w1, T* __restrict__ out, int C, int planes) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;     // one (block pair, lane, e)
    const int KS = C / 32;
    const long long total = (long long)(4 * C / 32) * KS * 2 * 512;
    if (i >= total) return;
    const int e = (int)(i & 7), lane = (int)((i >> 3) & 63);
    long long q = i >> 9;
    const int n = (int)(q & 1); q >>= 1;
    const int ks = (int)(q % KS);
    const int j = (int)(q / KS);
    const float v = w1[(long long)(32 * j + 16 * n + (lane & 15)) * C + 32 * ks + 8 * (lane >> 4) + e];
    const T h = (T)v;
    const long long o = ((((long long)j * KS + ks) * 2 + n) * planes << 9) + lane * 8 + e;
    out[o] = h;
    if (planes == 2) out[o + 512] = (T)(v - (float)h);
}

template <class T>
__global__ void prep_mlp_w2_kernel(const float* __restrict__ w2, T* __restrict__ out, int C, int planes) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int CF = C / 16;
    const long long total = (long long)(4 * C / 32) * CF * 512;
    if (i >= total) return;
    const int e = (int)(i & 7), lane = (int)((i >> 3) & 63);
    const long long q = i >> 9;
    const int c = (int)(q % CF);
    const int j = (int)(q / CF);
    const int col = perm8_col(16 * c + (lane & 15));
    const int hid = 32 * j + 16 * (e >> 2) + 4 * (lane >> 4) + (e & 3);
    const float v = w2[(long long)col * (4 * C) + hid];
    const T h = (T)v;
    const long long o = ((((long long)j * CF + c) * planes) << 9) + lane * 8 + e;
    out[o] = h;
    if (planes == 2) out[o + 512] = (T)(v - (float)h);
}

template <class T>
hipError_t prep_mlp_weights(const float* w1, const float* w2, T* w1f, T* w2f, int C, hipStream_t s, int planes) {
    if ((C != 192 && C != 384) || (planes != 1 && planes != 2)) return hipErrorInvalidValue;
    const long long t1 = (long long)(4 * C / 32) * (C / 32) * 2 * 512, t2 = (long long)(4 * C / 32) * (C / 16) * 512;
    hipLaunchKernelGGL((prep_mlp_w1_kernel<T>), dim3((unsigned)((t1 + 255) / 256)), dim3(256), 0, s, w1, w1f, C, planes);
    hipLaunchKernelGGL((prep_mlp_w2_kernel<T>), dim3((unsigned)((t2 + 255) / 256)), dim3(256), 0, s, w2, w2f, C, planes);
    return hipGetLastError();
}
template hipError_t prep_mlp_weights<bf16>(const float*, const float*, bf16*, bf16*, int, hipStream_t, int);
template hipError_t prep_mlp_weights<f16>(const float*, const float*, f16*, f16*, int, hipStream_t, int);

template <class T, class S>
static hipError_t launch_fused_mlp(const MlpArgs<T>& a, hipStream_t s) {
    auto kern = fused_mlp_kernel<T, S>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, S::SMEM);
    if (e != hipSuccess) return e;
    const unsigned grid = (unsigned)((a.M + S::BM - 1) / S::BM);
    if (grid == 0) return hipSuccess;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(S::THREADS), S::SMEM, s, a);
    return hipGetLastError();
}

template <class P>
hipError_t op_mlp_fused(const Geom& g, const BlockW<typename P::T>& b, int res, typename P::T* Xs, const Work<P>& wk, hipStream_t s) {
    typedef typename P::T T;
    static_assert(P::NA == 2 && P::NW == 2, "the fused MLP is the 3-term path");
    MlpArgs<T> a{Xs, wk.xs_plane[res], g.ntok[res], b.w1f, b.w2f, b.fc1_b, b.fc2_b, b.n2_g, b.n2_b, 1e-5f};
    if (a.M % 16 != 0) return hipErrorInvalidValue;
    // Measured at 721x1440 (ms per launch, C = 192 / C = 384; tools/mlp_variants.sh): the defaults are the schedules in which TWO
    // waves share a SIMD, so that one wave's GELU, LDS-DMA issue and barrier waits run under the other's MFMAs --
    //   C = 192: two independent 4-wave workgroups per CU, 32 tokens per wave ........ 0.748   (one 4-wave workgroup, 64 tokens per wave: 0.808;
    //            one 8-wave workgroup: 0.764 -- its waves reach the barriers, and therefore the GELU, together)
    //   C = 384: one 8-wave workgroup, 16 tokens per wave (LDS holds one 107 KB set) .. 0.695   (4 waves x 32 tokens, one per SIMD: 0.761)
    // The skewed schedule (GELU of chunk j spliced into fc1 of chunk j + 1) needs 16 more registers than a wave has and spills.
    static const int variant = [] { const char* v = getenv("SKP_MLP_VARIANT"); return v ? atoi(v) : 0; }();
    if (res == 0) {
        switch (variant) {
            case 1: return launch_fused_mlp<T, MlpShape<192, 2, 8, 3, false>>(a, s);
            case 2: return launch_fused_mlp<T, MlpShape<192, 4, 4, 3, false>>(a, s);
            case 3: return launch_fused_mlp<T, MlpShape<192, 4, 4, 2, true>>(a, s);
            default: return launch_fused_mlp<T, MlpShape<192, 2, 4, 3, false, 4, 2>>(a, s);
        }
    }
    switch (variant) {
        case 2: return launch_fused_mlp<T, MlpShape<384, 2, 4, 3, false>>(a, s);
        case 3: return launch_fused_mlp<T, MlpShape<384, 2, 4, 2, true>>(a, s);
        default: return launch_fused_mlp<T, MlpShape<384, 1, 8, 3, false>>(a, s);
    }
}
template hipError_t op_mlp_fused<PrecBF16x3>(const Geom&, const BlockW<bf16>&, int, bf16*, const Work<PrecBF16x3>&, hipStream_t);
template hipError_t op_mlp_fused<PrecF16x3>(const Geom&, const BlockW<f16>&, int, f16*, const Work<PrecF16x3>&, hipStream_t);

}  // namespace skp
