// Fused GEMM epilogues for gemm.h.  An epilogue receives the wave's accumulator fragments and owns
// everything between "sum over k" and HBM: bias, GELU, LayerNorm + residual, window reverse /
// un-roll / crop, pixel shuffle, QKV head split (+ V transpose), ConvTranspose scatter + de-normalise.
//
// Swapped order (default):  acc[a][b][r] = C[m = m0w + 16a + (lane&15)][n' = n0w + 16b + 4(lane>>4) + r]
// Un-swapped order:         acc[a][b][r] = C[m = m0w + 16a + 4(lane>>4) + r][n' = n0w + 16b + (lane&15)]
// n' is the PREPARED weight row; with the perm8 row order (common.h; every GEMM but PatchRecovery) the output column is
// n = perm8_col(n'), i.e. in swapped order the fragment pair (2bp, 2bp+1) of a lane holds the 8 consecutive columns
// n0w + 32bp + 8(lane>>4) + [0..7]  (acc[a][2bp][0..3] then acc[a][2bp+1][0..3]).
// Rule for every epilogue (measured): on gfx950 stores share the VMEM counter with loads, so a load that
// sits between two stores makes hipcc wait `vmcnt(0)` -- i.e. for the write acknowledgement of every store issued so far.
// All global loads of an epilogue are therefore issued BEFORE its first store, and the small per-column tables
// (bias, gamma, beta, mean/std) are staged into LDS by init() at kernel start and read back with ds_read.
#pragma once
#include "common.h"

namespace skp {

// keeps a value materialised HERE: stops LLVM from sinking the bias add into the masked store branches
__device__ __forceinline__ void pin(f32x4& v) { asm volatile("" : "+v"(v)); }

// the 8 consecutive columns of fragment pair bp (perm8)
__device__ __forceinline__ void add8(f32x4& lo, f32x4& hi, const float4& b0, const float4& b1) {
    lo[0] += b0.x; lo[1] += b0.y; lo[2] += b0.z; lo[3] += b0.w;
    hi[0] += b1.x; hi[1] += b1.y; hi[2] += b1.z; hi[3] += b1.w;
}

constexpr int kEpiReduceBytes = 4096;    // LayerNorm cross-wave reductions
constexpr int kEpiTableBytes = 8192;     // per-column tables staged by init()
constexpr int kEpiScratch = kEpiReduceBytes + kEpiTableBytes;


// ---- bias + store as residual-stream planes (PatchEmbedding, DownSample, UpSample.linear2) ---- //
// The residual stream has no fp32 copy: it lives as hi/lo 16-bit planes in the blocked operand layout (always two
// planes = 16 significand bits for bf16, 22 for fp16), which is what both its producers and its consumers want.
template <class T>
struct EpStorePlanes {
    static constexpr bool kDualOrder = false;
    template <class TC> __device__ __forceinline__ void init(char*, int, int) const {}
    const float* bias;      // nullable
    int ld, row_off;
    T* out;                 // hi/lo planes, blocked layout
    long long plane;
    template <class TC, bool SWAP>
    __device__ __forceinline__ void run(f32x4 (&acc)[TC::FM][TC::FN], int m0w, int n0w, int lane, int, int, char*, int M, int N, int) const {
        static_assert(SWAP && TC::FN % 2 == 0, "swapped order, fragment pairs");
        const int lm = lane & 15, l8 = (lane >> 4) * 8;
        // bias is added to every accumulator in straight-line code BEFORE the (masked, hence branchy) store loop: a loaded
        // register first used inside a branch makes hipcc emit vmcnt(0) there, which also waits for every earlier store
#pragma unroll
        for (int bp = 0; bp < TC::FN / 2; ++bp) {
            const int n = n0w + bp * 32 + l8;
            const float* bsrc = bias + (n < N ? n : 0);
            const float4 b0 = (bias != nullptr) ? *reinterpret_cast<const float4*>(bsrc) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 b1 = (bias != nullptr) ? *reinterpret_cast<const float4*>(bsrc + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int a = 0; a < TC::FM; ++a) { add8(acc[a][2 * bp], acc[a][2 * bp + 1], b0, b1); pin(acc[a][2 * bp]); pin(acc[a][2 * bp + 1]); }
        }
#pragma unroll
        for (int a = 0; a < TC::FM; ++a) {
            const int m = m0w + a * 16 + lm;
            if (m >= M) continue;
#pragma unroll
            for (int bp = 0; bp < TC::FN / 2; ++bp) {
                const int n = n0w + bp * 32 + l8;
                if (n >= N) continue;
                const f32x4 &x = acc[a][2 * bp], &y = acc[a][2 * bp + 1];
                const float vv[8] = {x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
                store8_planes<T, 2>(out + blk_off(m + row_off, n, ld), plane, vv);
            }
        }
    }
};

// ---- bias + GELU -> activation store (MLP fc1) ------------------------------- //
template <class T, int NPL>
struct EpGelu {
    static constexpr bool kDualOrder = false;
    template <class TC> __device__ __forceinline__ void init(char*, int, int) const {}
    T* out;                 // hi/lo planes
    long long plane;
    const float* bias;
    int ld;
    template <class TC, bool SWAP>
    __device__ __forceinline__ void run(f32x4 (&acc)[TC::FM][TC::FN], int m0w, int n0w, int lane, int, int, char*, int M, int N, int) const {
        static_assert(SWAP && TC::FN % 2 == 0, "swapped order, fragment pairs");
        const int lm = lane & 15, l8 = (lane >> 4) * 8;
#pragma unroll
        for (int bp = 0; bp < TC::FN / 2; ++bp) {      // straight-line bias add first (see EpStorePlanes)
            const int n = n0w + bp * 32 + l8;
            const float* bsrc = bias + (n < N ? n : 0);
            const float4 b0 = *reinterpret_cast<const float4*>(bsrc), b1 = *reinterpret_cast<const float4*>(bsrc + 4);
#pragma unroll
            for (int a = 0; a < TC::FM; ++a) { add8(acc[a][2 * bp], acc[a][2 * bp + 1], b0, b1); pin(acc[a][2 * bp]); pin(acc[a][2 * bp + 1]); }
        }
#pragma unroll
        for (int a = 0; a < TC::FM; ++a) {
            const int m = m0w + a * 16 + lm;
            if (m >= M) continue;
#pragma unroll
            for (int bp = 0; bp < TC::FN / 2; ++bp) {
                const int n = n0w + bp * 32 + l8;
                if (n >= N) continue;
                const f32x4 &x = acc[a][2 * bp], &y = acc[a][2 * bp + 1];
                float v[8];
                gelu_erf8(x, y, v);
                store8_planes<T, NPL>(out + blk_off(m, n, ld), plane, v);
            }
        }
    }
};

// ---- QKV head split: Q (scaled), K as [win][head][144][32]; V transposed [win][head][32][144] ---- //
// Q/K tiles run in swapped order (4 consecutive d per lane), V tiles in un-swapped order (4 consecutive
// tokens per lane) so that the attention kernel reads every MFMA fragment as contiguous bytes.
template <class T, int NPL>
struct EpQKV {
    static constexpr bool kDualOrder = true;
    template <class TC> __device__ __forceinline__ void init(char*, int, int) const {}
    T* q; T* k; T* vt;          // hi planes; lo plane at + plane
    long long plane;
    const float* bias;          // [3C]
    int C, heads;
    float scale;
    __device__ __forceinline__ bool unswapped(int n0) const { return n0 >= 2 * C; }
    template <class TC, bool SWAP>
    __device__ __forceinline__ void run(f32x4 (&acc)[TC::FM][TC::FN], int m0w, int n0w, int lane, int, int, char*, int M, int N, int) const {
        static_assert(TC::FN % 2 == 0 && TC::FM % 2 == 0, "fragment pairs");
        const int l15 = lane & 15, l4 = (lane >> 4) * 4, l8 = (lane >> 4) * 8;
        if constexpr (SWAP) {
#pragma unroll
            for (int bp = 0; bp < TC::FN / 2; ++bp) {      // straight-line bias add first (see EpStorePlanes)
                const int n = n0w + bp * 32 + l8;
                const float* bsrc = bias + (n < N ? n : 0);
                const float4 b0 = *reinterpret_cast<const float4*>(bsrc), b1 = *reinterpret_cast<const float4*>(bsrc + 4);
#pragma unroll
                for (int a = 0; a < TC::FM; ++a) { add8(acc[a][2 * bp], acc[a][2 * bp + 1], b0, b1); pin(acc[a][2 * bp]); pin(acc[a][2 * bp + 1]); }
            }
#pragma unroll
            for (int a = 0; a < TC::FM; ++a) {
                const int m = m0w + a * 16 + l15;
                if (m >= M) continue;
                const int win = m / WIN_TOKENS, t = m - win * WIN_TOKENS;
#pragma unroll
                for (int bp = 0; bp < TC::FN / 2; ++bp) {
                    const int n = n0w + bp * 32 + l8;       // 8 consecutive d of one head (C % 32 == 0)
                    if (n >= N) continue;
                    const int which = n >= C ? 1 : 0;
                    const int c = n - which * C;
                    const int head = c >> 5, d = c & 31;
                    const float s = which == 0 ? scale : 1.0f;
                    const f32x4 &x = acc[a][2 * bp], &y = acc[a][2 * bp + 1];
                    const float v[8] = {x[0] * s, x[1] * s, x[2] * s, x[3] * s, y[0] * s, y[1] * s, y[2] * s, y[3] * s};
                    T* dst = (which == 0 ? q : k) + (((long long)win * heads + head) * WIN_TOKENS + t) * HEAD_DIM + d;
                    store8_planes<T, NPL>(dst, plane, v);
                }
            }
        } else {
#pragma unroll
            for (int b = 0; b < TC::FN; ++b) {
                const int n = perm8_col(n0w + b * 16 + l15);
                const float bb = bias[n < N ? n : 0];
#pragma unroll
                for (int a = 0; a < TC::FM; ++a) { acc[a][b][0] += bb; acc[a][b][1] += bb; acc[a][b][2] += bb; acc[a][b][3] += bb; pin(acc[a][b]); }
            }
#pragma unroll
            for (int ap = 0; ap < TC::FM / 2; ++ap) {
                // the tile's A rows are in perm8 order (gemm_dma.h): fragment pair (2ap, 2ap+1) = 8 consecutive tokens
                const int m = m0w + ap * 32 + 2 * l4;
                if (m >= M) continue;
                const int win = m / WIN_TOKENS, t = m - win * WIN_TOKENS;     // 144 % 8 == 0: the 8 tokens share a window
#pragma unroll
                for (int b = 0; b < TC::FN; ++b) {
                    const int n = perm8_col(n0w + b * 16 + l15);
                    if (n >= N) continue;
                    const int c = n - 2 * C;
                    const int head = c >> 5, d = c & 31;
                    const f32x4 &x = acc[2 * ap][b], &y = acc[2 * ap + 1][b];
                    const float v[8] = {x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
                    store8_planes<T, NPL>(vt + (((long long)win * heads + head) * HEAD_DIM + d) * WIN_TOKENS + t, plane, v);
                }
            }
        }
    }
};

// ---- row LayerNorm over the tile's BN columns + (residual add | activation store) ---- //
// Requires the block to own complete LayerNorm groups: BN == group width (N == BN for proj/fc2,
// N == 4*BN for UpSample where each n-tile is one pixel-shuffle quadrant).
struct RowMapIndexed {          // proj: window reverse + un-roll + crop via the window table; fc2: identity
    const int* idx;             // nullable
    __device__ __forceinline__ long long dest(int m, int) const { return idx ? idx[m] : m; }
};
struct RowMapPixelShuffle {     // UpSample: (z,h,w) of the coarse grid, quadrant = n-tile -> fine token, cropped
    int H1, W1, H2, W2;
    __device__ __forceinline__ long long dest(int m, int ntile) const {
        const int hw = H2 * W2;
        const int z = m / hw, rem = m - z * hw;
        const int h = rem / W2, w = rem - h * W2;
        const int hf = 2 * h + (ntile >> 1), wf = 2 * w + (ntile & 1);
        if (hf >= H1 || wf >= W1) return -1;
        return ((long long)z * H1 + hf) * W1 + wf;
    }
};
struct f32x8 { float4 a, b; };
template <class T>
struct SinkResidual {           // xs[dest][c..c+7] += y on the hi/lo planes of the residual stream (16-byte loads / stores)
    T* xs;
    long long plane;
    static constexpr bool kLoads = true;
    __device__ __forceinline__ f32x8 load(long long row, int ld, int c) const {
        const T* p = xs + blk_off(row, c, ld);
        const typename OpT<T>::v8 h = as_v8<T>(*reinterpret_cast<const uint4*>(p));
        const typename OpT<T>::v8 l = as_v8<T>(*reinterpret_cast<const uint4*>(p + plane));
        f32x8 o;
        o.a = make_float4((float)h[0] + (float)l[0], (float)h[1] + (float)l[1], (float)h[2] + (float)l[2], (float)h[3] + (float)l[3]);
        o.b = make_float4((float)h[4] + (float)l[4], (float)h[5] + (float)l[5], (float)h[6] + (float)l[6], (float)h[7] + (float)l[7]);
        return o;
    }
    __device__ __forceinline__ void put(long long row, int ld, int c, const float (&y)[8], const f32x8& old) const {
        const float v[8] = {old.a.x + y[0], old.a.y + y[1], old.a.z + y[2], old.a.w + y[3],
                            old.b.x + y[4], old.b.y + y[5], old.b.z + y[6], old.b.w + y[7]};
        store8_planes<T, 2>(xs + blk_off(row, c, ld), plane, v);
    }
};
template <class T, int NPL>
struct SinkStore {              // out[dest][c..c+7] = y as hi/lo planes
    T* out;
    long long plane;
    static constexpr bool kLoads = false;
    __device__ __forceinline__ f32x8 load(long long, int, int) const { return f32x8{}; }
    __device__ __forceinline__ void put(long long row, int ld, int c, const float (&y)[8], const f32x8&) const {
        store8_planes<T, NPL>(out + blk_off(row, c, ld), plane, y);
    }
};

// GROUP > 1 (= the wave tile's FM): the tile's rows come in groups -- fragment a of a wave tile holds member a of the 16 groups
// (m0w / (16 FM)) * 16 + (lane & 15) -- and what is stored is the SUM of the members' LayerNorm outputs, one row per group
// (GraphCast's mesh->grid edges: the three edges into a grid node, summed by the receiver; graphcast_ops.hip).
template <class RowMap, class Sink, int GROUP = 1>
struct EpLayerNorm {
    static constexpr bool kDualOrder = false;
    // gamma | beta | bias of this tile's BN columns -> LDS (read back with ds_read: no VMEM load between the stores)
    template <class TC> __device__ __forceinline__ void init(char* tab, int tid, int n0) const {
        float* t = reinterpret_cast<float*>(tab);
        for (int i = tid; i < TC::BN; i += TC::THREADS) {
            t[i] = gamma[i];
            t[TC::BN + i] = beta[i];
            t[2 * TC::BN + i] = bias != nullptr ? bias[n0 + i] : 0.f;
        }
    }
    RowMap map;
    Sink sink;
    const float* bias;       // nullable, [N]
    const float* gamma;      // [BN]
    const float* beta;       // [BN]
    float eps;
    int n_groups = 0;        // GROUP > 1: number of groups (= output rows)
    template <class TC, bool SWAP>
    __device__ __forceinline__ void run(f32x4 (&acc)[TC::FM][TC::FN], int m0w, int n0w, int lane, int wm, int wn, char* smem, int M, int N, int ntile) const {
        static_assert(SWAP && TC::FN % 2 == 0, "swapped order, fragment pairs");
        static_assert(GROUP == 1 || (GROUP == TC::FM && !Sink::kLoads), "grouped rows: one member per fragment, no residual");
        constexpr int FM = TC::FM, FN = TC::FN, FP = TC::FN / 2, WN = TC::WN, BM = TC::BM, BN = TC::BN;
        const int l15 = lane & 15, l8 = (lane >> 4) * 8;
        const int nloc0 = n0w - ntile * BN;          // column of this wave inside the LN group
        float* red = reinterpret_cast<float*>(smem);  // [2][BM][WN]
        const float* tab = reinterpret_cast<const float*>(smem + kEpiReduceBytes);   // gamma | beta | bias (init())
        static_assert(3 * BN * 4 <= kEpiTableBytes && 2 * BM * WN * 4 <= kEpiReduceBytes, "epilogue LDS scratch");
#pragma unroll
        for (int bp = 0; bp < FP; ++bp) {
            const float* bsrc = tab + 2 * BN + nloc0 + bp * 32 + l8;
            const float4 b0 = *reinterpret_cast<const float4*>(bsrc), b1 = *reinterpret_cast<const float4*>(bsrc + 4);
#pragma unroll
            for (int a = 0; a < FM; ++a) add8(acc[a][2 * bp], acc[a][2 * bp + 1], b0, b1);
        }
        // destination rows first (the table lookups overlap the LayerNorm reductions below); rows that do not exist
        // (window padding / crop / M tail) are redirected to row 0 for the loads and skipped for the stores
        long long drow[FM];
        bool ok[FM];
#pragma unroll
        for (int a = 0; a < FM; ++a) {
            const int m = m0w + a * 16 + l15;
            const long long r = map.dest(m < M ? m : 0, ntile);     // branch-free: the FM table lookups go out back to back
            ok[a] = m < M && r >= 0;
            drow[a] = ok[a] ? r : 0;
        }
        // residual loads: the first PRE 16-row groups go out now (their round trip hides under the LayerNorm
        // reductions); the last one is issued after group 0 has been stored, when its accumulators are dead --
        // all FM*FN loads at once spill next to the FM*FN accumulators
        constexpr int PRE = FM < 3 ? FM : 3;
        f32x8 old[Sink::kLoads ? FM : 1][Sink::kLoads ? FP : 1];
        if constexpr (Sink::kLoads) {
#pragma unroll
            for (int a = 0; a < PRE; ++a)
#pragma unroll
                for (int bp = 0; bp < FP; ++bp) old[a][bp] = sink.load(drow[a], BN, nloc0 + bp * 32 + l8);
        }
        float mean[FM], rstd[FM];
        // pass 1: mean
#pragma unroll
        for (int a = 0; a < FM; ++a) {
            float s = 0.f;
#pragma unroll
            for (int b = 0; b < FN; ++b) s += (acc[a][b][0] + acc[a][b][1]) + (acc[a][b][2] + acc[a][b][3]);
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            if (lane < 16) red[(wm * TC::WTM + a * 16 + lane) * WN + wn] = s;
        }
        __syncthreads();
#pragma unroll
        for (int a = 0; a < FM; ++a) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < WN; ++w) s += red[(wm * TC::WTM + a * 16 + l15) * WN + w];
            mean[a] = s * (1.0f / BN);
        }
        // pass 2: centred second moment
        float* red2 = red + BM * WN;
#pragma unroll
        for (int a = 0; a < FM; ++a) {
            float s = 0.f;
#pragma unroll
            for (int b = 0; b < FN; ++b) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float d = acc[a][b][r] - mean[a]; s += d * d; }
            }
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            if (lane < 16) red2[(wm * TC::WTM + a * 16 + lane) * WN + wn] = s;
        }
        __syncthreads();
#pragma unroll
        for (int a = 0; a < FM; ++a) {
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < WN; ++w) s += red2[(wm * TC::WTM + a * 16 + l15) * WN + w];
            rstd[a] = rsqrtf(s * (1.0f / BN) + eps);
        }
        if constexpr (GROUP > 1) {
            const long long grow = (long long)(m0w / (16 * FM)) * 16 + l15;
#pragma unroll
            for (int bp = 0; bp < FP; ++bp) {
                const int c = nloc0 + bp * 32 + l8;
                const float4 g0 = *reinterpret_cast<const float4*>(tab + c), g1 = *reinterpret_cast<const float4*>(tab + c + 4);
                const float4 e0 = *reinterpret_cast<const float4*>(tab + BN + c), e1 = *reinterpret_cast<const float4*>(tab + BN + c + 4);
                float y[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int a = 0; a < FM; ++a) {
                    const f32x4 &x = acc[a][2 * bp], &z = acc[a][2 * bp + 1];
                    const float mu = mean[a], rs = rstd[a];
                    y[0] += (x[0] - mu) * rs * g0.x + e0.x; y[1] += (x[1] - mu) * rs * g0.y + e0.y;
                    y[2] += (x[2] - mu) * rs * g0.z + e0.z; y[3] += (x[3] - mu) * rs * g0.w + e0.w;
                    y[4] += (z[0] - mu) * rs * g1.x + e1.x; y[5] += (z[1] - mu) * rs * g1.y + e1.y;
                    y[6] += (z[2] - mu) * rs * g1.z + e1.z; y[7] += (z[3] - mu) * rs * g1.w + e1.w;
                }
                if (grow < n_groups) sink.put(grow, BN, c, y, old[0][0]);
            }
            return;
        }
#pragma unroll
        for (int a = 0; a < FM; ++a) {
#pragma unroll
            for (int bp = 0; bp < FP; ++bp) {
                const int c = nloc0 + bp * 32 + l8;
                const float4 g0 = *reinterpret_cast<const float4*>(tab + c), g1 = *reinterpret_cast<const float4*>(tab + c + 4);
                const float4 e0 = *reinterpret_cast<const float4*>(tab + BN + c), e1 = *reinterpret_cast<const float4*>(tab + BN + c + 4);
                const f32x4 &x = acc[a][2 * bp], &z = acc[a][2 * bp + 1];
                const float mu = mean[a], rs = rstd[a];
                const float y[8] = {(x[0] - mu) * rs * g0.x + e0.x, (x[1] - mu) * rs * g0.y + e0.y, (x[2] - mu) * rs * g0.z + e0.z, (x[3] - mu) * rs * g0.w + e0.w,
                                    (z[0] - mu) * rs * g1.x + e1.x, (z[1] - mu) * rs * g1.y + e1.y, (z[2] - mu) * rs * g1.z + e1.z, (z[3] - mu) * rs * g1.w + e1.w};
                if (ok[a]) sink.put(drow[a], BN, c, y, Sink::kLoads ? old[a][bp] : old[0][0]);
            }
            if constexpr (Sink::kLoads) {
                if (a + PRE < FM) {
#pragma unroll
                    for (int bp = 0; bp < FP; ++bp) old[a + PRE][bp] = sink.load(drow[a + PRE], BN, nloc0 + bp * 32 + l8);
                }
            }
        }
    }
};

// ---- PatchRecovery: ConvTranspose scatter + crop + de-normalise --------------- //
// Upper air: token (zt,h,w), n = ((v*2+dz)*4+dh)*4+dw -> state[v*13 + 2zt+dz][4h+dh-top][4w+dw]
// Surface:   token (h,w),    n = (v*4+dh)*4+dw       -> state[surf0+v][4h+dh-top][4w+dw]
struct EpRecover {
    static constexpr bool kDualOrder = false;
    // mean[0..68] | std[72..140] | bias[144..151] -> LDS
    template <class TC> __device__ __forceinline__ void init(char* tab, int tid, int) const {
        float* t = reinterpret_cast<float*>(tab);
        if (tid < 69) { t[tid] = mean[tid]; t[72 + tid] = std[tid]; }
        if (tid < 8) t[144 + tid] = tid < (surface ? 4 : 5) ? bias[tid] : 0.f;
    }
    float* state;            // [69][n_lat][n_lon]
    const float* bias;       // [n_vars]
    const float* mean;       // [69]
    const float* std;        // [69]
    int n_lat, n_lon, lat_top, H1, W1, n_levels, surface, surf0;
    template <class TC, bool SWAP>
    __device__ __forceinline__ void run(f32x4 (&acc)[TC::FM][TC::FN], int m0w, int n0w, int lane, int, int, char* smem, int M, int N, int) const {
        static_assert(SWAP, "swapped order only");
        const int l15 = lane & 15, l4 = (lane >> 4) * 4;
        const int hw = H1 * W1;
        const float* tab = reinterpret_cast<const float*>(smem + kEpiReduceBytes);
#pragma unroll
        for (int a = 0; a < TC::FM; ++a) {
            const int m = m0w + a * 16 + l15;
            if (m >= M) continue;
            const int zt = m / hw, rem = m - zt * hw;
            const int h = rem / W1, w = rem - h * W1;
#pragma unroll
            for (int b = 0; b < TC::FN; ++b) {
                const int n = n0w + b * 16 + l4;      // dw = 0..3 are this lane's 4 values
                if (n >= N) continue;
                const int dh = (n >> 2) & 3;
                int ch, v;
                if (surface) { v = n >> 4; ch = surf0 + v; }
                else {
                    v = n >> 5;
                    const int level = 2 * zt + ((n >> 4) & 1);
                    if (level >= n_levels) continue;
                    ch = v * n_levels + level;
                }
                const int lat = 4 * h + dh - lat_top;
                if (lat < 0 || lat >= n_lat) continue;
                const float bb = tab[144 + v], sd = tab[72 + ch], mu = tab[ch];
                const float4 o = make_float4((acc[a][b][0] + bb) * sd + mu, (acc[a][b][1] + bb) * sd + mu,
                                             (acc[a][b][2] + bb) * sd + mu, (acc[a][b][3] + bb) * sd + mu);
                *reinterpret_cast<float4*>(state + ((long long)ch * n_lat + lat) * n_lon + 4 * w) = o;
            }
        }
    }
};

}  // namespace skp
