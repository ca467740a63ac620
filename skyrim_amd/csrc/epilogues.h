// Fused GEMM epilogues for gemm.h.  An epilogue receives the wave's accumulator fragments and owns
// everything between "sum over k" and HBM: bias, GELU, LayerNorm + residual, window reverse /
// un-roll / crop, pixel shuffle, QKV head split (+ V transpose), ConvTranspose scatter + de-normalise.
//
// Swapped order (default):  acc[a][b][r] = C[m = m0w + 16a + (lane&15)][n = n0w + 16b + 4(lane>>4) + r]
// Un-swapped order:         acc[a][b][r] = C[m = m0w + 16a + 4(lane>>4) + r][n = n0w + 16b + (lane&15)]
// Rule for every epilogue (measured): on gfx950 stores share the VMEM counter with loads, so a load that
// sits between two stores makes hipcc wait `vmcnt(0)` -- i.e. for the write acknowledgement of every store issued so far.
// All global loads of an epilogue are therefore issued BEFORE its first store, and the small per-column tables
// (bias, gamma, beta, mean/std) are staged into LDS by init() at kernel start and read back with ds_read.
#pragma once
#include "common.h"

namespace skp {

// keeps a value materialised HERE: stops LLVM from sinking the bias add into the masked store branches
__device__ __forceinline__ void pin(f32x4& v) { asm volatile("" : "+v"(v)); }

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
        static_assert(SWAP, "swapped order only");
        const int lm = lane & 15, ln = (lane >> 4) * 4;
        // bias is added to every accumulator in straight-line code BEFORE the (masked, hence branchy) store loop: a loaded
        // register first used inside a branch makes hipcc emit vmcnt(0) there, which also waits for every earlier store
#pragma unroll
        for (int b = 0; b < TC::FN; ++b) {
            const int n = n0w + b * 16 + ln;
            const float4 bb = (bias != nullptr) ? *reinterpret_cast<const float4*>(bias + (n < N ? n : 0)) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int a = 0; a < TC::FM; ++a) { acc[a][b][0] += bb.x; acc[a][b][1] += bb.y; acc[a][b][2] += bb.z; acc[a][b][3] += bb.w; pin(acc[a][b]); }
        }
#pragma unroll
        for (int a = 0; a < TC::FM; ++a) {
            const int m = m0w + a * 16 + lm;
            if (m >= M) continue;
#pragma unroll
            for (int b = 0; b < TC::FN; ++b) {
                const int n = n0w + b * 16 + ln;
                if (n >= N) continue;
                const float4 v = make_float4(acc[a][b][0], acc[a][b][1], acc[a][b][2], acc[a][b][3]);
                const float vv[4] = {v.x, v.y, v.z, v.w};
                store4_planes<T, 2>(out + blk_off(m + row_off, n, ld), plane, vv);
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
        static_assert(SWAP, "swapped order only");
        const int lm = lane & 15, ln = (lane >> 4) * 4;
#pragma unroll
        for (int b = 0; b < TC::FN; ++b) {      // straight-line bias add first (see EpStoreF32)
            const int n = n0w + b * 16 + ln;
            const float4 bb = *reinterpret_cast<const float4*>(bias + (n < N ? n : 0));
#pragma unroll
            for (int a = 0; a < TC::FM; ++a) { acc[a][b][0] += bb.x; acc[a][b][1] += bb.y; acc[a][b][2] += bb.z; acc[a][b][3] += bb.w; pin(acc[a][b]); }
        }
#pragma unroll
        for (int a = 0; a < TC::FM; ++a) {
            const int m = m0w + a * 16 + lm;
            if (m >= M) continue;
#pragma unroll
            for (int b = 0; b < TC::FN; ++b) {
                const int n = n0w + b * 16 + ln;
                if (n >= N) continue;
                const float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
#ifdef SKP_DEBUG_NOGELU
                float v[4] = {acc[a][b][0] + bb.x, acc[a][b][1] + bb.y, acc[a][b][2] + bb.z, acc[a][b][3] + bb.w};
#else
                float v[4] = {gelu_erf(acc[a][b][0] + bb.x), gelu_erf(acc[a][b][1] + bb.y),
                              gelu_erf(acc[a][b][2] + bb.z), gelu_erf(acc[a][b][3] + bb.w)};
#endif
#ifdef SKP_DEBUG_NOSTORE
                if (v[0] == 123.456f)
#endif
                store4_planes<T, NPL>(out + blk_off(m, n, ld), plane, v);
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
        const int l15 = lane & 15, l4 = (lane >> 4) * 4;
        if constexpr (SWAP) {
#pragma unroll
            for (int b = 0; b < TC::FN; ++b) {      // straight-line bias add first (see EpStoreF32)
                const int n = n0w + b * 16 + l4;
                const float4 bb = *reinterpret_cast<const float4*>(bias + (n < N ? n : 0));
#pragma unroll
                for (int a = 0; a < TC::FM; ++a) { acc[a][b][0] += bb.x; acc[a][b][1] += bb.y; acc[a][b][2] += bb.z; acc[a][b][3] += bb.w; pin(acc[a][b]); }
            }
#pragma unroll
            for (int a = 0; a < TC::FM; ++a) {
                const int m = m0w + a * 16 + l15;
                if (m >= M) continue;
                const int win = m / WIN_TOKENS, t = m - win * WIN_TOKENS;
#pragma unroll
                for (int b = 0; b < TC::FN; ++b) {
                    const int n = n0w + b * 16 + l4;
                    if (n >= N) continue;
                    const int which = n >= C ? 1 : 0;
                    const int c = n - which * C;
                    const int head = c >> 5, d = c & 31;
                    const float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
                    const float s = which == 0 ? scale : 1.0f;
                    const float v[4] = {(acc[a][b][0] + bb.x) * s, (acc[a][b][1] + bb.y) * s,
                                        (acc[a][b][2] + bb.z) * s, (acc[a][b][3] + bb.w) * s};
                    uint2 o[NPL];
                    split4<T, NPL>(v, o);
                    T* dst = (which == 0 ? q : k) + (((long long)win * heads + head) * WIN_TOKENS + t) * HEAD_DIM + d;
#pragma unroll
                    for (int p = 0; p < NPL; ++p) *reinterpret_cast<uint2*>(dst + p * plane) = o[p];
                }
            }
        } else {
#pragma unroll
            for (int b = 0; b < TC::FN; ++b) {
                const int n = n0w + b * 16 + l15;
                const float bb = bias[n < N ? n : 0];
#pragma unroll
                for (int a = 0; a < TC::FM; ++a) { acc[a][b][0] += bb; acc[a][b][1] += bb; acc[a][b][2] += bb; acc[a][b][3] += bb; pin(acc[a][b]); }
            }
#pragma unroll
            for (int a = 0; a < TC::FM; ++a) {
                const int m = m0w + a * 16 + l4;
                if (m >= M) continue;
                const int win = m / WIN_TOKENS, t = m - win * WIN_TOKENS;
#pragma unroll
                for (int b = 0; b < TC::FN; ++b) {
                    const int n = n0w + b * 16 + l15;
                    if (n >= N) continue;
                    const int c = n - 2 * C;
                    const int head = c >> 5, d = c & 31;
                    const float bb = 0.f;
                    const float v[4] = {acc[a][b][0] + bb, acc[a][b][1] + bb, acc[a][b][2] + bb, acc[a][b][3] + bb};
                    uint2 o[NPL];
                    split4<T, NPL>(v, o);
                    T* dst = vt + (((long long)win * heads + head) * HEAD_DIM + d) * WIN_TOKENS + t;
#pragma unroll
                    for (int p = 0; p < NPL; ++p) *reinterpret_cast<uint2*>(dst + p * plane) = o[p];
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
template <class T>
struct SinkResidual {           // xs[dest][c..c+3] += y on the hi/lo planes of the residual stream
    T* xs;
    long long plane;
    static constexpr bool kLoads = true;
    __device__ __forceinline__ float4 load(long long row, int ld, int c) const {
        typedef T t4 __attribute__((ext_vector_type(4)));
        const T* p = xs + blk_off(row, c, ld);
        const t4 h = __builtin_bit_cast(t4, *reinterpret_cast<const uint2*>(p));
        const t4 l = __builtin_bit_cast(t4, *reinterpret_cast<const uint2*>(p + plane));
        return make_float4((float)h[0] + (float)l[0], (float)h[1] + (float)l[1], (float)h[2] + (float)l[2], (float)h[3] + (float)l[3]);
    }
    __device__ __forceinline__ void put(long long row, int ld, int c, const float (&y)[4], const float4& old) const {
        const float v[4] = {old.x + y[0], old.y + y[1], old.z + y[2], old.w + y[3]};
        store4_planes<T, 2>(xs + blk_off(row, c, ld), plane, v);
    }
};
template <class T, int NPL>
struct SinkStore {              // out[dest][c..c+3] = y as hi/lo planes
    T* out;
    long long plane;
    static constexpr bool kLoads = false;
    __device__ __forceinline__ float4 load(long long, int, int) const { return make_float4(0.f, 0.f, 0.f, 0.f); }
    __device__ __forceinline__ void put(long long row, int ld, int c, const float (&y)[4], const float4&) const {
        store4_planes<T, NPL>(out + blk_off(row, c, ld), plane, y);
    }
};

template <class RowMap, class Sink>
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
    template <class TC, bool SWAP>
    __device__ __forceinline__ void run(f32x4 (&acc)[TC::FM][TC::FN], int m0w, int n0w, int lane, int wm, int wn, char* smem, int M, int N, int ntile) const {
        static_assert(SWAP, "swapped order only");
        constexpr int FM = TC::FM, FN = TC::FN, WN = TC::WN, BM = TC::BM, BN = TC::BN;
        const int l15 = lane & 15, l4 = (lane >> 4) * 4;
        const int nloc0 = n0w - ntile * BN;          // column of this wave inside the LN group
        float* red = reinterpret_cast<float*>(smem);  // [2][BM][WN]
        const float* tab = reinterpret_cast<const float*>(smem + kEpiReduceBytes);   // gamma | beta | bias (init())
        static_assert(3 * BN * 4 <= kEpiTableBytes && 2 * BM * WN * 4 <= kEpiReduceBytes, "epilogue LDS scratch");
#pragma unroll
        for (int b = 0; b < FN; ++b) {
            const float4 bb = *reinterpret_cast<const float4*>(tab + 2 * BN + nloc0 + b * 16 + l4);
#pragma unroll
            for (int a = 0; a < FM; ++a) { acc[a][b][0] += bb.x; acc[a][b][1] += bb.y; acc[a][b][2] += bb.z; acc[a][b][3] += bb.w; }
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
        float4 old[Sink::kLoads ? FM : 1][Sink::kLoads ? FN : 1];
        if constexpr (Sink::kLoads) {
#pragma unroll
            for (int a = 0; a < PRE; ++a)
#pragma unroll
                for (int b = 0; b < FN; ++b) old[a][b] = sink.load(drow[a], BN, nloc0 + b * 16 + l4);
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
#pragma unroll
        for (int a = 0; a < FM; ++a) {
#pragma unroll
            for (int b = 0; b < FN; ++b) {
                const int c = nloc0 + b * 16 + l4;
                const float4 g = *reinterpret_cast<const float4*>(tab + c);
                const float4 be = *reinterpret_cast<const float4*>(tab + BN + c);
                const float y[4] = {(acc[a][b][0] - mean[a]) * rstd[a] * g.x + be.x, (acc[a][b][1] - mean[a]) * rstd[a] * g.y + be.y,
                                    (acc[a][b][2] - mean[a]) * rstd[a] * g.z + be.z, (acc[a][b][3] - mean[a]) * rstd[a] * g.w + be.w};
                if (ok[a]) sink.put(drow[a], BN, c, y, Sink::kLoads ? old[a][b] : old[0][0]);
            }
            if constexpr (Sink::kLoads) {
                if (a + PRE < FM) {
#pragma unroll
                    for (int b = 0; b < FN; ++b) old[a + PRE][b] = sink.load(drow[a + PRE], BN, nloc0 + b * 16 + l4);
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
