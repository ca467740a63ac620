// GraphCast interaction-network updates as ONE kernel each (include/skyrim_graphcast.h: skgc_edge_update, skgc_node_mlp).
//
// An edge update of the reference's typed graph network (/root/reference/skyrim/core/models/graphcast.py:118 -> DeepMind's
// InteractionNetwork) is   e' = e + LayerNorm(W2 swish(W1 concat(e, v_s[send], v_r[recv]) + b1) + b2),   agg[recv] += (e' - e).
// By distributivity W1 concat(...) = e W_e^T + (v_s W_s^T)[send] + (v_r W_r^T)[recv]: the node terms are computed once per NODE by a plain
// GEMM; here the rest runs for a tile of 128 edge rows without leaving the CU:
//
//   * a wavefront owns 2 x 16 rows.  Their 512 input columns sit in REGISTERS as MFMA B-operand fragments (one fp16 plane, 16-byte
//     loads of the blocked layout: one wave instruction = one contiguous 1 KiB block); the weights stream through LDS in FRAGMENT
//     order by LDS-DMA (two stages, one barrier per stage).  The second Linear keeps fp16 hi/lo weight planes (two MFMA terms
//     W_hi h + W_lo h); the edge part of the first Linear takes one or two planes (skgc_edge_desc::w1_planes).  Measured on the oracle at
//     production width and depth (tools/graphcast_numerics.py, error relative to the predicted increment, bar 1e-3): activations of the
//     edge MLPs as one fp16 plane 1.7e-4; W_e as one plane +1.4e-4; the second Linear's weights as one plane would cost 4.5e-4.
//   * phase 1: the first Linear in chunks of 32 hidden units.  The accumulators START from the gathered node terms (stored in "pos"
//     column order, skyrim_amd/graphcast/fused.py: a lane's 8 values are 32 contiguous bytes; the gathers of chunk j + 1 fly during
//     chunk j) and END, after swish and the fp16 rounding, as the second Linear's B-operand fragments -- in swapped order
//     (D^T = W X^T) a lane's accumulators ARE its k-slots of the next GEMM (same trick as fused_mlp.hip).  The swish of chunk j - 1,
//     the gathers of chunk j + 1 and the LDS-DMA requests of the next stage are spliced between the MFMAs of chunk j.
//   * phase 2: the second Linear, 16 k-steps x 32 output fragments into 2 x 128 accumulator registers; perm8 weight rows make a
//     fragment pair 8 consecutive output columns.
//   * MFMA order: a v_mfma_f32_16x16x32_f16 that accumulates into the result of an earlier one issues ~47 clocks after it (measured:
//     chains at distance 2 run at 23.7 clocks per MFMA, the pipe's 16 need distance >= 3), so every step interleaves FOUR accumulator
//     chains (2 row groups x 2 fragments) before it returns to the first.
//   * epilogue: LayerNorm from the accumulators (in-lane sums + two shuffles), the residual update written back in the blocked
//     layout, and the RECEIVER SUM: the normalised rows go through LDS (two halves of 256 columns over the weight stages, 260-float
//     row stride: conflict-free both ways), one thread per column walks the tile's 128 rows in order and writes a run's sum where the
//     receiver changes -- sequential, deterministic, no atomics, no second kernel.  The host packs the rows so that a run never crosses
//     a 128-row tile unless it is longer than a tile; the pieces of such a run go to a side buffer and skgc_segment_fixup adds them.
//
// The `static` form (grid->mesh, mesh->grid: edge latents that do not depend on the input) has no first Linear at all: its phase 1 is
// swish(prepared term + gathered node terms).  The node form (skgc_node_mlp) keeps fp32 node latents exact: hi/lo fragments made on
// the fly from fp32 rows, three MFMA terms, one 16-row group per wave, concatenated sources taken one after the other.
//
// gfx950 only.  One workgroup (4 waves, one per SIMD, 512 registers) per CU; grid = tiles.
#include <cstdlib>
#include "../../include/skyrim_graphcast.h"
#include "gemm_dma.h"

namespace skp {

__device__ __forceinline__ float swish_f(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }

constexpr int FZ_RD = 2;           // steps of weight fragments in flight per wave (register ring)
// The ONE build-time switch of the library's measurement paths (never set by the Makefile): -DSKP_PROBES=<bits> compiles timing probes whose
// RESULTS ARE WRONG -- & 1: no weight DMA after a kernel's first stages; & 2: no MFMAs in the chunk loops (tools/gc_edge_probe.py).
#ifndef SKP_PROBES
#define SKP_PROBES 0
#endif
__device__ __forceinline__ f32x4 fz_mfma(const uint4& a, const OpT<f16>::v8& b, const f32x4& c) {
    if (SKP_PROBES & 2) return c;
    return OpT<f16>::mfma(as_v8<f16>(a), b, c);
}
constexpr int FZ_L = 512, FZ_KS = 16, FZ_CF = 32, FZ_NCH = 16, FZ_TILE = 128;
constexpr int FZ_STAGE = 64 * 1024;
constexpr int FZ_YLD = 260;                                   // row stride (floats) of the epilogue's exchange buffer: 260 mod 32 = 4
constexpr int FZ_YBUF = FZ_TILE * FZ_YLD * 4;                 // 133 120 bytes: over both weight stages and 5 KiB more
constexpr int FZ_TAB = 5 * FZ_L;                              // b2 | gamma | beta | b1 (node form) | receivers of the tile (edge form)
constexpr int FZ_SMEM = FZ_YBUF + FZ_TAB * 4;
static_assert(FZ_YBUF >= 2 * FZ_STAGE && FZ_SMEM <= 160 * 1024, "LDS");

struct EdgeArgs {
    const f16* e_in;        // FC1: edge latents; static: prepared first-Linear term ("pos" columns).  Blocked fp16 [rows][512]
    f16* e_out;             // nullable (FC1 only): e_in + LayerNorm(...), blocked fp16; may alias e_in
    const float* term[2];   // gathered node terms, fp32 rows in "pos" column order
    const int* idx[2];      // node of every packed row (< 0: padding row)
    long long ld[2];
    const int* recv;        // receiver of every packed row (< 0: padding row); rows sorted by receiver
    const f16* w1f;         // FC1: first Linear (edge part), fragment order, 1 or 2 planes
    const f16* w2f;         // second Linear, fragment order, hi/lo planes
    const float *b2, *gamma, *beta;
    float* agg;             // [nodes][512] receiver sums
    float* heads;           // [tiles][512] continuation pieces (runs longer than a tile)
    float eps;
    long long* probe;       // nullable: [tiles][8] shader clocks of wave 0 at the phase boundaries (measurement only)
    int xcd_order;          // 1: XCD x works through the contiguous tile range x (SKGC_XCD_TILE_ORDER=1); 0: launch order
};

// NB consecutive 1 KiB fragments per step through a ring of RD steps, read RD - 1 steps ahead of their MFMAs; the scheduling barrier pins
// the reads in program order (fused_mlp.hip: ld_pair)
template <int NB>
__device__ __forceinline__ void fz_ld(const char* p, uint4 (&w)[NB]) {
#pragma unroll
    for (int i = 0; i < NB; ++i) w[i] = *reinterpret_cast<const uint4*>(p + i * 1024);
    __builtin_amdgcn_sched_barrier(0);
}
template <int NSTEP, int NB, int RD, class Body>
__device__ __forceinline__ void fz_steps(const char* st, Body&& body) {
    uint4 ring[RD][NB];
#pragma unroll
    for (int s = 0; s < RD - 1 && s < NSTEP; ++s) fz_ld<NB>(st + s * NB * 1024, ring[s % RD]);
#pragma unroll
    for (int s = 0; s < NSTEP; ++s) {
        if (s + RD - 1 < NSTEP) fz_ld<NB>(st + (s + RD - 1) * NB * 1024, ring[(s + RD - 1) % RD]);
        body(s, ring[s % RD]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// piece k of a stage's LDS-DMA: 1 KiB block wave + 4 k of `src` -> the same block of the LDS stage at `dst`.  Pieces are requested ONE AT
// A TIME between MFMAs: issuing one costs the wave ~60-85 clocks (measured: 16 in a burst add 1.4 k clocks per stage to a wave that has
// the SIMD to itself), which the matrix pipe covers only when it has work queued.
__device__ __forceinline__ void fz_piece(const f16* src, unsigned dst, int k, int wave, int lane, bool first = false) {
    if ((SKP_PROBES & 1) && !first) return;
    const int q = wave + 4 * k;
    glds16(src + (q << 9) + lane * 8, dst + (unsigned)(q << 10));
}

__device__ __forceinline__ f16x8 fz_pack8(const f32x4& a, const f32x4& b) {
    f16x8 h;
#pragma unroll
    for (int i = 0; i < 4; ++i) { h[i] = (f16)a[i]; h[4 + i] = (f16)b[i]; }
    return h;
}

// 8 fp32 -> fp16 hi / lo fragments.  The hi plane is made opaque before the lo plane is derived from it: left alone, hipcc converts twice
// (v_cvt_pk_f16_f32 for the stored plane, v_cvt_f16_f32 for the one the remainder is taken from) and on gfx950 the two do not always agree,
// which shows as a full fp16 ulp of error in single elements (measured: 17 of 2 M hidden activations, all within one fp32 ulp of an fp16
// grid point).
__device__ __forceinline__ void fz_split8(const float (&v)[8], f16x8& hi, f16x8& lo) {
#pragma unroll
    for (int i = 0; i < 8; ++i) hi[i] = (f16)v[i];
    asm volatile("" : "+v"(hi));
#pragma unroll
    for (int i = 0; i < 8; ++i) lo[i] = (f16)(v[i] - (float)hi[i]);
}

// a[j] with a wave-uniform RUNTIME j: a uniform 16-way branch around a register copy, so that the chunk loops stay rolled (fully unrolled
// they are ~200 KB of code against a 64 KB instruction cache: measured 1.6x slower in the MFMA loops) and the array stays in registers
#define FZ_CASES(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
// (the empty asm in every case keeps the cases apart: without it LLVM folds the switch back into a dynamically indexed array, which lives in
// scratch memory)
template <class V>
__device__ __forceinline__ V fz_pick(const V (&a)[16], int j) {
    V r = a[0];
    switch (j) {
#define X(i) case i: r = a[i]; asm volatile("" : "+v"(r)); break;
        FZ_CASES(X)
#undef X
        default: break;
    }
    return r;
}
template <class V>
__device__ __forceinline__ void fz_put(V (&a)[16], int j, V v) {
    switch (j) {
#define X(i) case i: asm volatile("" : "+v"(v)); a[i] = v; break;
        FZ_CASES(X)
#undef X
        default: break;
    }
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

// W1P: planes of the first Linear's weights (FC1 only).  Fragment order of a chunk of 32 hidden units: [ks][n][plane] -- a step = one k-step
// against both 16-unit halves, i.e. four accumulator chains with the two row groups.
template <bool FC1, int NT, int W1P>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
edge_update_kernel(const EdgeArgs a) {
    constexpr int FM = 2, KS = FZ_KS, CF = FZ_CF, NCH = FZ_NCH, RD = FZ_RD;
    constexpr int W1_CHUNK = 16 * 2 * W1P * 512;                      // elements of one chunk of the first Linear: 16 k-steps x 2 halves x planes KiB
    constexpr int W1_PIECES = 16 * 2 * W1P / 4;                       // DMA pieces per wave and chunk (8 or 16)
    typedef typename OpT<f16>::v8 v8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* tab = reinterpret_cast<float*>(smem + FZ_YBUF);
    int* rcv_l = reinterpret_cast<int*>(tab + 4 * FZ_L);              // [0] = receiver of the row before the tile, [1 + r] = of row r, [129] = -3
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds_base = (unsigned)(size_t)smem;
    const char* lrd = smem + lane * 16;
    // The tile of this workgroup: launch order, or (opt-in, SKGC_XCD_TILE_ORDER=1) XCD x working through the contiguous tile range x -- workgroups
    // reach the 8 XCDs round-robin, so that keeps neighbouring tiles (receiver-sorted edges) on one L2.  Measured at full size: the processor
    // kernel fetches 8 % less (1.41 -> 1.30 GB per launch; the sender rows still miss: the multi-mesh numbers its nodes level by level, not by
    // position) and runs 0.8 % SLOWER (22.8 -> 23.0 ms per step over the 16 layers), so launch order stays the default.
    int tile_id = blockIdx.x;
    if (a.xcd_order) {
        const int T = gridDim.x, q = T >> 3, r = T & 7, x = tile_id & 7, slot = tile_id >> 3;
        tile_id = x < r ? x * (q + 1) + slot : r * (q + 1) + (x - r) * q + slot;
    }
    const long long tile0 = (long long)tile_id * FZ_TILE;
    auto stamp = [&](int k) {
        if (a.probe != nullptr && tid == 0) a.probe[(long long)blockIdx.x * 8 + k] = (long long)__builtin_amdgcn_s_memtime();
    };
    stamp(0);

    // ---- prologue: first weight stages, tables, rows ------------------------------------------------------------------------------ //
    if constexpr (FC1) {
#pragma unroll
        for (int k = 0; k < W1_PIECES; ++k) fz_piece(a.w1f, lds_base, k, wave, lane, true);
    } else {
#pragma unroll
        for (int k = 0; k < 16; ++k) fz_piece(a.w2f, lds_base, k, wave, lane, true);
#pragma unroll
        for (int k = 0; k < 16; ++k) fz_piece(a.w2f + (FZ_STAGE / 2), lds_base + FZ_STAGE, k, wave, lane, true);
    }
    for (int i = tid; i < FZ_L; i += 256) { tab[i] = a.b2[i]; tab[FZ_L + i] = a.gamma[i]; tab[2 * FZ_L + i] = a.beta[i]; }
    if (tid < FZ_TILE) rcv_l[1 + tid] = a.recv[tile0 + tid];
    if (tid == FZ_TILE) rcv_l[0] = tile0 > 0 ? a.recv[tile0 - 1] : -2;
    if (tid == FZ_TILE + 1) rcv_l[1 + FZ_TILE] = -3;

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
    if constexpr (FC1) {
#pragma unroll
        for (int t = 0; t < FM; ++t)
#pragma unroll
            for (int k = 0; k < NCH; ++k) hh[t][k] = f16x8{};
    }
    f32x4 pre[FM][2];                                                 // pre-activations of the chunk whose swish is in flight
    static_assert(FM == 2 && NCH == 16, "fz_pick / fz_put and the step bodies are written for two groups of 16 chunks");

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
        constexpr int NG = FM * (NT > 0 ? NT : 1) * 2;                 // gathered 16-byte pieces per lane and chunk
        // two sets of gathered pieces: the requests of chunk j + 2 fly during chunk j + 1 (a gather out of the 168 MB node-term table takes
        // ~2 us, longer than a chunk), so the chunk loop is unrolled by two and a chunk's top waits with vmcnt(NG), not vmcnt(0)
        f32x4 gatA[NG], gatB[NG];
        auto gather1 = [&](f32x4 (&gat)[NG], int j, int i) {          // piece i = (t, s, half)
            if (NT == 0 || i >= FM * NT * 2) return;
            const int t = i / (NT * 2), s = (i / 2) % (NT > 0 ? NT : 1), h = i & 1;
            gat[i] = *reinterpret_cast<const f32x4*>(tp[t][s] + 32 * j + 4 * h);
        };
#pragma unroll
        for (int i = 0; i < NG; ++i) gather1(gatA, 0, i);
#pragma unroll
        for (int i = 0; i < NG; ++i) gather1(gatB, 1, i);
        // consumed once here so that hipcc's vmcnt waits for these loads sit BEFORE the loop (inside it they would also wait for the
        // LDS-DMA in flight, which the compiler's counter bookkeeping does not know about; fused_mlp.hip)
#pragma unroll
        for (int t = 0; t < FM; ++t)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(xh[t][ks]));
        stamp(1);

        f32x4 hacc[FM][2];
#pragma unroll
        for (int t = 0; t < FM; ++t) { hacc[t][0] = f32x4{0.f, 0.f, 0.f, 0.f}; hacc[t][1] = hacc[t][0]; }
        // one chunk; gat: the set holding chunk j's gathered pieces, refilled with chunk j + 2's
        auto chunk = [&](int j, f32x4 (&gat)[NG]) {
            // everything but the newest NG requests (the other set's, issued last in the previous chunk) has landed: W1(j) and this set
            if constexpr (NT > 0) {
                if (j == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                else __builtin_amdgcn_s_waitcnt(0x0F70 | NG);          // (the builtin: the compiler's own bookkeeping follows it, see the static form)
                asm volatile("" ::: "memory");
#pragma unroll
                for (int i = 0; i < NG; ++i) asm volatile("" : "+v"(gat[i]));
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();                           // W1(j) landed in stage j & 1; every wave is done with the other stage
            // the MFMA-only arrays live in the accumulation half of the register file (MFMA reads its operands from there directly):
            // left alone, hipcc homes part of them in the low half and spills VALU values to scratch around the chunk loop
#pragma unroll
            for (int t = 0; t < FM; ++t) {
#pragma unroll
                for (int k = 0; k < KS; ++k) { asm volatile("" : "+a"(xh[t][k])); asm volatile("" : "+a"(hh[t][k])); }
            }
#pragma unroll
            for (int t = 0; t < FM; ++t) {             // chunk j - 1's pre-activations (j = 0: zeros, result dropped)
                pre[t][0] = hacc[t][0]; pre[t][1] = hacc[t][1];
                f32x4 lo = {0.f, 0.f, 0.f, 0.f}, hi = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int s = 0; s < NT; ++s) {
                    lo += gat[(t * NT + s) * 2];
                    hi += gat[(t * NT + s) * 2 + 1];
                }
                hacc[t][0] = lo; hacc[t][1] = hi;
            }
            asm volatile("" ::: "memory");
            const bool more = j + 1 < NCH;
            const f16* nsrc = more ? a.w1f + (long long)(j + 1) * W1_CHUNK : a.w2f;      // chunk 15 reads stage 1: stage 0 is free for W2(0)
            const unsigned ndst = lds_base + (more ? ((j + 1) & 1) * FZ_STAGE : 0);
            const int jg = j + 2 < NCH ? j + 2 : NCH - 1;                                 // past the end: chunk 15 again, dropped
            f16x8 done[FM];
            // step ks: W1 rows 32 j + 16 n + [0, 16) for n = 0, 1 against k-step ks.  Between the MFMAs: the next stage's DMA pieces in the
            // first half of the chunk, THEN chunk j + 2's gathers (so that they are the newest requests at the next top), and the swish of
            // chunk j - 1
            fz_steps<KS, 2 * W1P, RD>(lrd + (j & 1) * FZ_STAGE, [&](int ks, const uint4 (&w)[2 * W1P]) {
                if constexpr (W1P == 2) {
                    hacc[0][0] = fz_mfma(w[1], xh[0][ks], hacc[0][0]);
                    hacc[1][0] = fz_mfma(w[1], xh[1][ks], hacc[1][0]);
                    hacc[0][1] = fz_mfma(w[3], xh[0][ks], hacc[0][1]);
                    hacc[1][1] = fz_mfma(w[3], xh[1][ks], hacc[1][1]);
                }
                hacc[0][0] = fz_mfma(w[0], xh[0][ks], hacc[0][0]);
                hacc[1][0] = fz_mfma(w[0], xh[1][ks], hacc[1][0]);
                hacc[0][1] = fz_mfma(w[W1P], xh[0][ks], hacc[0][1]);
                hacc[1][1] = fz_mfma(w[W1P], xh[1][ks], hacc[1][1]);
                if (ks < 8) {
                    fz_piece(nsrc, ndst, ks, wave, lane);
                    if (W1P == 2 || !more) fz_piece(nsrc, ndst, 8 + ks, wave, lane);     // 16 pieces: two planes of W1, or W2(0)
                } else if (ks - 8 < NG) {
                    gather1(gat, jg, ks - 8);          // unconditional: a load inside a branch makes hipcc wait vmcnt(0) at the join
                }
                if (ks == 2) swish4(pre[0][0]);
                if (ks == 3) swish4(pre[0][1]);
                if (ks == 4) swish4(pre[1][0]);
                if (ks == 5) swish4(pre[1][1]);
                if (ks == 6) done[0] = fz_pack8(pre[0][0], pre[0][1]);
                if (ks == 7) done[1] = fz_pack8(pre[1][0], pre[1][1]);
            });
            fz_put(hh[0], j - 1, done[0]);
            fz_put(hh[1], j - 1, done[1]);
        };
#pragma unroll 1
        for (int j = 0; j < NCH; j += 2) {
            chunk(j, gatA);
            chunk(j + 1, gatB);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the last (dropped) gathers
#pragma unroll
        for (int i = 0; i < NG; ++i) { asm volatile("" : "+v"(gatA[i])); asm volatile("" : "+v"(gatB[i])); }
#pragma unroll
        for (int t = 0; t < FM; ++t) { pre[t][0] = hacc[t][0]; pre[t][1] = hacc[t][1]; }
    }

    // ---- phase 2: second Linear ----------------------------------------------------------------------------------------------------- //
    stamp(2);
    f32x4 yacc[FM][CF];
#pragma unroll
    for (int t = 0; t < FM; ++t)
#pragma unroll
        for (int c = 0; c < CF; ++c) yacc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};
    // step q of a chunk: output fragments 2 q, 2 q + 1 (hi, lo planes each) -- four accumulator chains
    auto fc2_step = [&](int q, const uint4 (&w)[4], const f16x8& h0, const f16x8& h1) {
        yacc[0][2 * q] = fz_mfma(w[1], h0, yacc[0][2 * q]);
        yacc[1][2 * q] = fz_mfma(w[1], h1, yacc[1][2 * q]);
        yacc[0][2 * q + 1] = fz_mfma(w[3], h0, yacc[0][2 * q + 1]);
        yacc[1][2 * q + 1] = fz_mfma(w[3], h1, yacc[1][2 * q + 1]);
        yacc[0][2 * q] = fz_mfma(w[0], h0, yacc[0][2 * q]);
        yacc[1][2 * q] = fz_mfma(w[0], h1, yacc[1][2 * q]);
        yacc[0][2 * q + 1] = fz_mfma(w[2], h0, yacc[0][2 * q + 1]);
        yacc[1][2 * q + 1] = fz_mfma(w[2], h1, yacc[1][2 * q + 1]);
    };
    if constexpr (FC1) {
        // the last chunk's swish has nothing left to hide under: do it here
#pragma unroll
        for (int t = 0; t < FM; ++t) { swish4(pre[t][0]); swish4(pre[t][1]); hh[t][NCH - 1] = fz_pack8(pre[t][0], pre[t][1]); }
#pragma unroll
        for (int t = 0; t < FM; ++t)
#pragma unroll
            for (int k = 0; k < NCH; ++k) asm volatile("" : "+a"(hh[t][k]));
#pragma unroll 1
        for (int j = 0; j < NCH; ++j) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                           // W2(j) landed in stage j & 1; every wave is done with the other stage
            const bool more = j + 1 < NCH;
            const f16* nsrc = a.w2f + (long long)(j + 1) * (FZ_STAGE / 2);
            const unsigned ndst = lds_base + ((j + 1) & 1) * FZ_STAGE;
            const f16x8 h0 = fz_pick(hh[0], j), h1 = fz_pick(hh[1], j);
            fz_steps<CF / 2, 4, RD>(lrd + (j & 1) * FZ_STAGE, [&](int q, const uint4 (&w)[4]) {
                fc2_step(q, w, h0, h1);
                if (more && q < 8) { fz_piece(nsrc, ndst, q, wave, lane); fz_piece(nsrc, ndst, 8 + q, wave, lane); }
            });
        }
    } else {
        // static form: the pre-activation of chunk j = prepared term (fp16, "pos" columns: the lane's 16 bytes of block (row block, j)) +
        // gathered node terms.  Its pieces are requested TWO chunks ahead (they fly during chunk j - 2), summed at the top of chunk j - 1
        // and go through swish between the MFMAs of chunk j - 1: the whole first phase hides under the second Linear.
        constexpr int NG = FM * (1 + 2 * NT);                           // pieces per lane and chunk: per row group one fp16 piece + 2 per term
        // two sets, as in the first form: chunk j + 2's pieces are requested LAST in chunk j (after the DMA pieces), summed at the top of
        // chunk j + 1 -- which waits with vmcnt(NG): everything but chunk j + 3's requests -- and go through swish during chunk j + 1
        f32x4 gatA[NG], gatB[NG];                                       // (the fp16 piece is a 16-byte load as well)
        auto piece = [&](f32x4 (&gat)[NG], int j, int i) {
            const int t = i / (1 + 2 * NT), k = i % (1 + 2 * NT);
            if (k == 0) gat[i] = *reinterpret_cast<const f32x4*>(a.e_in + ((((rb0 + t) * KS) + j) << 9) + l15 * 32 + g * 8);
            else gat[i] = *reinterpret_cast<const f32x4*>(tp[t][(k - 1) >> 1] + 32 * j + 4 * ((k - 1) & 1));
        };
        auto sum_pre = [&](const f32x4 (&gat)[NG]) {                    // gat -> pre (frees the set for the next requests)
#pragma unroll
            for (int t = 0; t < FM; ++t) {
                const f16x8 st = __builtin_bit_cast(f16x8, gat[t * (1 + 2 * NT)]);
                f32x4 lo = {(float)st[0], (float)st[1], (float)st[2], (float)st[3]}, hi = {(float)st[4], (float)st[5], (float)st[6], (float)st[7]};
#pragma unroll
                for (int s = 0; s < NT; ++s) { lo += gat[t * (1 + 2 * NT) + 1 + 2 * s]; hi += gat[t * (1 + 2 * NT) + 2 + 2 * s]; }
                pre[t][0] = lo; pre[t][1] = hi;
            }
        };
        static_assert(NG == 2 || NG == 6 || NG == 10, "vmcnt immediates below");
        f16x8 hc[FM], hn[FM];
#pragma unroll
        for (int i = 0; i < NG; ++i) piece(gatB, 0, i);
        sum_pre(gatB);
#pragma unroll
        for (int i = 0; i < NG; ++i) piece(gatA, 1, i);                 // chunk 1 -> set A, chunk 2 -> set B, ...
#pragma unroll
        for (int i = 0; i < NG; ++i) piece(gatB, 2, i);
#pragma unroll
        for (int t = 0; t < FM; ++t) { swish4(pre[t][0]); swish4(pre[t][1]); hc[t] = fz_pack8(pre[t][0], pre[t][1]); hn[t] = hc[t]; }
        // chunk j: MFMAs on hc = hidden activation of chunk j; `gat` holds chunk j + 1's pieces and is refilled with chunk j + 3's
        auto chunk = [&](int j, f32x4 (&gat)[NG]) {
            // a waitcnt the COMPILER sees (the builtin, not asm): its own bookkeeping then knows this set has landed and adds no vmcnt(0) of its
            // own at the loop header; the LDS-DMA requests it does not know about are older than the set that stays in flight
            __builtin_amdgcn_s_waitcnt(0x0F70 | NG);
            asm volatile("" ::: "memory");
#pragma unroll
            for (int i = 0; i < NG; ++i) asm volatile("" : "+v"(gat[i]));    // chunk j + 1's pieces have landed: to the compiler HERE
            __syncthreads();                           // W2(j) landed in stage j & 1; every wave is done with the other stage
            sum_pre(gat);                              // pre = pre-activation of chunk j + 1 (past the end: chunk 15 again, dropped)
            asm volatile("" ::: "memory");
            const bool more = j + 1 < NCH && j > 0;                       // W2(0) and W2(1) were requested up front
            const f16* nsrc = a.w2f + (long long)(j + 1) * (FZ_STAGE / 2);
            const unsigned ndst = lds_base + ((j + 1) & 1) * FZ_STAGE;
            const int jn = j + 3 < NCH ? j + 3 : NCH - 1;
            fz_steps<CF / 2, 4, RD>(lrd + (j & 1) * FZ_STAGE, [&](int q, const uint4 (&w)[4]) {
                fc2_step(q, w, hc[0], hc[1]);
                if (q < 8) {
                    if (more) { fz_piece(nsrc, ndst, q, wave, lane); fz_piece(nsrc, ndst, 8 + q, wave, lane); }
                } else {
                    if (2 * (q - 8) < NG) piece(gat, jn, 2 * (q - 8));
                    if (2 * (q - 8) + 1 < NG) piece(gat, jn, 2 * (q - 8) + 1);
                }
                if (q == 2) swish4(pre[0][0]);
                if (q == 3) swish4(pre[0][1]);
                if (q == 4) swish4(pre[1][0]);
                if (q == 5) swish4(pre[1][1]);
                if (q == 6) hn[0] = fz_pack8(pre[0][0], pre[0][1]);
                if (q == 7) hn[1] = fz_pack8(pre[1][0], pre[1][1]);
            });
            hc[0] = hn[0]; hc[1] = hn[1];
        };
        __builtin_amdgcn_s_waitcnt(0x0F70 | ((2 * NG) & 15) | (((2 * NG) >> 4) << 14));   // only the two sets are in flight at the loop's entry
#pragma unroll 1
        for (int j = 0; j < NCH; j += 2) {
            chunk(j, gatA);
            chunk(j + 1, gatB);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the last (dropped) requests
#pragma unroll
        for (int i = 0; i < NG; ++i) { asm volatile("" : "+v"(gatA[i])); asm volatile("" : "+v"(gatB[i])); }
    }

    // ---- epilogue: LayerNorm, residual update, receiver sum ---------------------------------------------------------------------- //
    stamp(3);
    if constexpr (FC1) {
        if (a.e_out != nullptr) {
#pragma unroll
            for (int t = 0; t < FM; ++t) {
                const long long off = (((rb0 + t) * KS) << 9) + l15 * 32 + g * 8;
                v8 xr[KS];
#pragma unroll
                for (int bp = 0; bp < KS; ++bp) xr[bp] = *reinterpret_cast<const v8*>(a.e_in + off + (bp << 9));
                fz_layer_norm(yacc[t], tab, g, a.eps);
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
        } else {
#pragma unroll
            for (int t = 0; t < FM; ++t) fz_layer_norm(yacc[t], tab, g, a.eps);
        }
    } else {
#pragma unroll
        for (int t = 0; t < FM; ++t) fz_layer_norm(yacc[t], tab, g, a.eps);
    }
    stamp(4);
    // receiver sums: the 128 normalised rows through LDS, 256 columns at a time; thread c walks column c down the rows in order
    float* ybuf = reinterpret_cast<float*>(smem);
    const int first = rcv_l[1];
    const bool tile_cont = first >= 0 && rcv_l[0] == first;          // the tile's first run continues the previous tile's last one
    // bit r of (ends_hi : ends_lo): the run of row r ends there (the next row has another receiver, or the tile ends)
    const int rv0 = rcv_l[1 + lane], rv1 = rcv_l[65 + lane];         // receivers of rows lane, 64 + lane: read back with v_readlane at a run's end
    const unsigned long long ends_lo = __ballot(rcv_l[2 + lane] != rv0), ends_hi = __ballot(rcv_l[66 + lane] != rv1);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        __syncthreads();                               // every wave is done with the weight stages / with the previous half
#pragma unroll
        for (int t = 0; t < FM; ++t) {
            float* dst = ybuf + (wave * 32 + t * 16 + l15) * FZ_YLD + 8 * g;
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const f32x4 &x = yacc[t][2 * (8 * half + b)], &z = yacc[t][2 * (8 * half + b) + 1];
                *reinterpret_cast<float4*>(dst + 32 * b) = make_float4(x[0], x[1], x[2], x[3]);
                *reinterpret_cast<float4*>(dst + 32 * b + 4) = make_float4(z[0], z[1], z[2], z[3]);
            }
        }
        __syncthreads();
        const float* col = ybuf + tid;
        float* const out_c = a.agg + 256 * half + tid;
        float* const head_c = a.heads + (long long)tile_id * FZ_L + 256 * half + tid;
        float acc = 0.f;
#pragma unroll 1
        for (int r0 = 0; r0 < FZ_TILE; r0 += 16) {
            float v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = col[(r0 + i) * FZ_YLD];          // 16 rows in flight; the run logic below is scalar
            const unsigned ends = (unsigned)((r0 < 64 ? ends_lo : ends_hi) >> (r0 & 63)) & 0xffffu;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                acc += v[i];
                if (ends & (1u << i)) {                    // the run ends with row r0 + i (uniform branch)
                    const int cur = __builtin_amdgcn_readlane(r0 < 64 ? rv0 : rv1, (r0 + i) & 63);
                    if (cur >= 0) {
                        if (tile_cont && cur == first) { if (a.heads) *head_c = acc; }      // heads == NULL: the caller vouched that no tile continues (header)
                        else out_c[(long long)cur * FZ_L] = acc;
                    }
                    acc = 0.f;
                }
            }
        }
    }
    stamp(5);
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
// 4-5e-4 of the predicted increment (tools/graphcast_numerics.py), so the kernel splits every fp32 operand into fp16 hi/lo fragments on the
// fly: A W^T ~ A_hi W_lo^T + A_lo W_hi^T + A_hi W_hi^T.  128-row tiles, 2 x 16 rows per wave.  The first Linear runs K-OUTER: the 512 hidden
// pre-activations of both row groups are the accumulators (256 registers), a 64 KiB LDS stage is ONE k-step of all 512 hidden units (fragment
// order [source][ks][n = 32 unit groups][plane], fused.py: prep_w1_fragments_kouter) and feeds 192 MFMAs; the operand's 32 columns of the next
// k-step are loaded and split while the current one computes; concatenated sources are further k-steps.  The second Linear walks W2 chunk by
// chunk once per row group (its LayerNorm needs a group's 512 outputs in registers, and two groups' outputs do not fit beside the hidden
// activations).  Round 5, 1 038 240 rows: 4.07 ms (one source) / 5.21 ms (two) against 4.71 / 6.76 ms for the form that held one row group's
// whole operand and walked the hidden units chunk by chunk (96 MFMAs per stage), 41 k rows: 0.24 / 0.32 against 0.22 / 0.32; a third form
// (hidden units in two halves, every stage against both row groups) measured 4.66 / 6.30 and 0.29 / 0.38 (docs/experiments.md).
struct NodeArgs {
    const float* src[2];    // fp32 rows, 512 columns each (concatenated along K)
    long long ld[2];
    const f16 *w1f, *w2f;   // fragment order, hi/lo planes: n_src x [512][512], [512][512]
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
    constexpr int KS = FZ_KS, CF = FZ_CF, NCH = FZ_NCH, RD = FZ_RD, FM = 2;
    typedef typename OpT<f16>::v8 v8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* tab = reinterpret_cast<float*>(smem + FZ_YBUF);            // b2 | gamma | beta | b1
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds_base = (unsigned)(size_t)smem;
    const char* lrd = smem + lane * 16;

#pragma unroll
    for (int k = 0; k < 16; ++k) fz_piece(a.w1f, lds_base, k, wave, lane, true);
    for (int i = tid; i < FZ_L; i += 256) { tab[i] = a.b2[i]; tab[FZ_L + i] = a.gamma[i]; tab[2 * FZ_L + i] = a.beta[i]; tab[3 * FZ_L + i] = a.b1[i]; }

    long long row[FM], rr[FM];
    bool live[FM];
#pragma unroll
    for (int t = 0; t < FM; ++t) {
        row[t] = (long long)blockIdx.x * FZ_TILE + wave * 32 + t * 16 + l15;
        live[t] = row[t] < a.rows;
        rr[t] = live[t] ? row[t] : a.rows - 1;
    }
    // hp[t][half][j]: first the accumulators of hidden units 32 j + 16 half + 4 g + r of row group t, then (after swish) the hi (half 0) / lo
    // (half 1) fragment of k-step j of the second Linear -- the same registers, 8 fp32 <-> 2 x 8 fp16 as in the first form
    f32x4 hp[FM][2][NCH];
#pragma unroll
    for (int t = 0; t < FM; ++t)
#pragma unroll
        for (int j = 0; j < NCH; ++j) { hp[t][0][j] = f32x4{0.f, 0.f, 0.f, 0.f}; hp[t][1][j] = f32x4{0.f, 0.f, 0.f, 0.f}; }

    // the operand's columns of the first k-step
    float4 u[FM][2];
#pragma unroll
    for (int t = 0; t < FM; ++t) {
        const float* p = a.src[0] + rr[t] * a.ld[0] + 8 * g;
        u[t][0] = *reinterpret_cast<const float4*>(p);
        u[t][1] = *reinterpret_cast<const float4*>(p + 4);
    }
    int stage = 0;
#pragma unroll 1
    for (int it = 0; it < NS * KS; ++it) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                               // this k-step's weights and operand columns landed; every wave is done with the other stage
        v8 xh[FM], xl[FM];
#pragma unroll
        for (int t = 0; t < FM; ++t) {
            const float v[8] = {u[t][0].x, u[t][0].y, u[t][0].z, u[t][0].w, u[t][1].x, u[t][1].y, u[t][1].z, u[t][1].w};
            fz_split8(v, xh[t], xl[t]);
        }
        const bool last = it + 1 == NS * KS;
        {                                              // the next k-step's columns fly under this one's MFMAs (unconditional: a load inside a
            const int nx = last ? it : it + 1;         //  branch makes hipcc wait vmcnt(0) at the join; the last k-step re-reads its own columns)
            const int s1 = NS > 1 && nx >= KS ? 1 : 0, k1 = nx - s1 * KS;
            const float* base = s1 ? a.src[NS - 1] : a.src[0];
            const long long ld = s1 ? a.ld[NS - 1] : a.ld[0];
#pragma unroll
            for (int t = 0; t < FM; ++t) {
                const float* p = base + rr[t] * ld + 32 * k1 + 8 * g;
                u[t][0] = *reinterpret_cast<const float4*>(p);
                u[t][1] = *reinterpret_cast<const float4*>(p + 4);
            }
        }
        const f16* nsrc = last ? a.w2f : a.w1f + (long long)(it + 1) * (FZ_STAGE / 2);
        const unsigned ndst = lds_base + (stage ^ 1) * FZ_STAGE;
        // step q: unit groups 2 q, 2 q + 1 (= chunk q, halves 0 / 1), fragments [n][plane]; four accumulator chains x three terms
        fz_steps<NCH, 4, RD>(lrd + stage * FZ_STAGE, [&](int q, const uint4 (&w)[4]) {
            hp[0][0][q] = fz_mfma(w[0], xh[0], hp[0][0][q]);
            hp[1][0][q] = fz_mfma(w[0], xh[1], hp[1][0][q]);
            hp[0][1][q] = fz_mfma(w[2], xh[0], hp[0][1][q]);
            hp[1][1][q] = fz_mfma(w[2], xh[1], hp[1][1][q]);
            if (q < 8) fz_piece(nsrc, ndst, 2 * q, wave, lane);
            hp[0][0][q] = fz_mfma(w[0], xl[0], hp[0][0][q]);
            hp[1][0][q] = fz_mfma(w[0], xl[1], hp[1][0][q]);
            hp[0][1][q] = fz_mfma(w[2], xl[0], hp[0][1][q]);
            hp[1][1][q] = fz_mfma(w[2], xl[1], hp[1][1][q]);
            if (q < 8) fz_piece(nsrc, ndst, 2 * q + 1, wave, lane);
            hp[0][0][q] = fz_mfma(w[1], xh[0], hp[0][0][q]);
            hp[1][0][q] = fz_mfma(w[1], xh[1], hp[1][0][q]);
            hp[0][1][q] = fz_mfma(w[3], xh[0], hp[0][1][q]);
            hp[1][1][q] = fz_mfma(w[3], xh[1], hp[1][1][q]);
        });
        stage ^= 1;
    }
    // + b1, swish, hi / lo split: the hidden fragments of both row groups (tab was filled before the first barrier)
#pragma unroll
    for (int t = 0; t < FM; ++t)
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const float4 b0 = *reinterpret_cast<const float4*>(tab + 3 * FZ_L + 32 * j + 4 * g), b1 = *reinterpret_cast<const float4*>(tab + 3 * FZ_L + 32 * j + 16 + 4 * g);
            const f32x4 lo = hp[t][0][j], hi = hp[t][1][j];
            const float v[8] = {swish_f(lo[0] + b0.x), swish_f(lo[1] + b0.y), swish_f(lo[2] + b0.z), swish_f(lo[3] + b0.w),
                                swish_f(hi[0] + b1.x), swish_f(hi[1] + b1.y), swish_f(hi[2] + b1.z), swish_f(hi[3] + b1.w)};
            f16x8 o0, o1;
            fz_split8(v, o0, o1);
            hp[t][0][j] = __builtin_bit_cast(f32x4, o0);
            hp[t][1][j] = __builtin_bit_cast(f32x4, o1);
        }

    // second Linear + LayerNorm + residual, one row group after the other (W2 streams through the stages once per group)
#pragma unroll
    for (int t = 0; t < FM; ++t) {
        f32x4 yacc[CF];
#pragma unroll
        for (int c = 0; c < CF; ++c) yacc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
        for (int j = 0; j < NCH; ++j) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const bool more = !(t == FM - 1 && j == NCH - 1);
            const f16* nsrc = j + 1 < NCH ? a.w2f + (long long)(j + 1) * (FZ_STAGE / 2) : a.w2f;      // (the next group starts over at chunk 0)
            const unsigned ndst = lds_base + (stage ^ 1) * FZ_STAGE;
            const v8 h = __builtin_bit_cast(v8, fz_pick(hp[t][0], j)), l = __builtin_bit_cast(v8, fz_pick(hp[t][1], j));
            fz_steps<CF / 4, 8, RD>(lrd + stage * FZ_STAGE, [&](int q, const uint4 (&w)[8]) {
#pragma unroll
                for (int i = 0; i < 4; ++i) yacc[4 * q + i] = fz_mfma(w[2 * i + 1], h, yacc[4 * q + i]);
                if (more && q < 4) { fz_piece(nsrc, ndst, 4 * q, wave, lane); fz_piece(nsrc, ndst, 4 * q + 1, wave, lane); }
#pragma unroll
                for (int i = 0; i < 4; ++i) yacc[4 * q + i] = fz_mfma(w[2 * i], l, yacc[4 * q + i]);
                if (more && q < 4) { fz_piece(nsrc, ndst, 4 * q + 2, wave, lane); fz_piece(nsrc, ndst, 4 * q + 3, wave, lane); }
#pragma unroll
                for (int i = 0; i < 4; ++i) yacc[4 * q + i] = fz_mfma(w[2 * i], h, yacc[4 * q + i]);
            });
            stage ^= 1;
        }
        float4 r0[FZ_KS], r1[FZ_KS];
        if (a.res != nullptr) {
            const float* p = a.res + rr[t] * a.ld_res + 8 * g;
#pragma unroll
            for (int bp = 0; bp < FZ_KS; ++bp) { r0[bp] = *reinterpret_cast<const float4*>(p + 32 * bp); r1[bp] = *reinterpret_cast<const float4*>(p + 32 * bp + 4); }
        } else {
#pragma unroll
            for (int bp = 0; bp < FZ_KS; ++bp) { r0[bp] = make_float4(0.f, 0.f, 0.f, 0.f); r1[bp] = r0[bp]; }
        }
        fz_layer_norm(yacc, tab, g, a.eps);
        if (live[t]) {
            float* dst = a.out + row[t] * a.ld_out + 8 * g;
#pragma unroll
            for (int bp = 0; bp < FZ_KS; ++bp) {
                const f32x4 &x = yacc[2 * bp], &z = yacc[2 * bp + 1];
                *reinterpret_cast<float4*>(dst + 32 * bp) = make_float4(r0[bp].x + x[0], r0[bp].y + x[1], r0[bp].z + x[2], r0[bp].w + x[3]);
                *reinterpret_cast<float4*>(dst + 32 * bp + 4) = make_float4(r1[bp].x + z[0], r1[bp].y + z[1], r1[bp].z + z[2], r1[bp].w + z[3]);
            }
        }
    }
}

}  // namespace skp

using namespace skp;

template <bool FC1, int NT, int W1P>
static int launch_edge(const EdgeArgs& a, long long tiles, hipStream_t st) {
    auto kern = edge_update_kernel<FC1, NT, W1P>;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, FZ_SMEM) != hipSuccess) return SKGC_E_HIP;
    hipLaunchKernelGGL(kern, dim3((unsigned)tiles), dim3(256), FZ_SMEM, st, a);
    return hipGetLastError() == hipSuccess ? 0 : SKGC_E_HIP;
}

extern "C" {

int skgc_edge_update(const skgc_edge_desc* d, void* stream) {
    if (!d || !d->e_in || !d->recv || !d->w2f || !d->b2 || !d->gamma || !d->beta || !d->agg || d->rows <= 0 || (d->rows % FZ_TILE) || d->n_term < 0 || d->n_term > 2 ||
        (d->has_fc1 && (!d->w1f || (d->w1_planes != 1 && d->w1_planes != 2))) || (!d->has_fc1 && d->e_out) || (reinterpret_cast<size_t>(d->e_in) & 15) ||
        (reinterpret_cast<size_t>(d->e_out) & 15))
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
    a.probe = d->probe;
    static const bool xcd_order = getenv("SKGC_XCD_TILE_ORDER") != nullptr;
    a.xcd_order = xcd_order ? 1 : 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (d->has_fc1) {
        if (d->w1_planes == 2) {
            if (d->n_term == 2) return launch_edge<true, 2, 2>(a, tiles, st);
            if (d->n_term == 1) return launch_edge<true, 1, 2>(a, tiles, st);
            return launch_edge<true, 0, 2>(a, tiles, st);
        }
        if (d->n_term == 2) return launch_edge<true, 2, 1>(a, tiles, st);
        if (d->n_term == 1) return launch_edge<true, 1, 1>(a, tiles, st);
        return launch_edge<true, 0, 1>(a, tiles, st);
    }
    if (d->n_term == 2) return launch_edge<false, 2, 1>(a, tiles, st);
    if (d->n_term == 1) return launch_edge<false, 1, 1>(a, tiles, st);
    return launch_edge<false, 0, 1>(a, tiles, st);
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
    const long long tiles = (d->rows + FZ_TILE - 1) / FZ_TILE;
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
