// Shared pieces of the strided fp32 GEMMs behind include/skyrim_sfno.h and include/skyrim_graphcast.h: the fp32 A-operand
// loader (strides, per-k affine, two sources along K), the strided epilogue (bias, residual before / after the activation,
// erf-GELU or swish) and the batched kernel wrapper around gemm.h's register-staged main loop.
#pragma once
#include <hip/hip_runtime.h>

#include "common.h"
#include "epilogues.h"
#include "gemm.h"
#include "launchers.h"

namespace skp {

// ---- A operand: fp32, strided ------------------------------------------------------------------ //
struct ALStrided {
    static constexpr bool kDirect = false;
    // rows contiguous and k strided (an NCHW activation read pixel-major: GraphCast's embedding, the un-fused SFNO encoder / decoder): stage
    // rows-first, like ALFast<false> (gemm.h: al_rows_first_t); decided per launch from the strides
    static constexpr bool kRowsFirst = true;
    __device__ __forceinline__ bool rows_first() const { return sm == 1 && sk != 1 && (a2 == nullptr || sk2 != 1); }
    const float* a;
    int M, K, m1;                 // row m -> (m / m1) * sm2 + (m % m1) * sm
    long long sm, sm2, sk;
    const float* kscale;          // optional per-k affine applied BEFORE the fp16 split: A'(m, k) = A(m, k) * kscale[k] + kshift[k]
    const float* kshift;          // (input normalisation: raw fields such as geopotential ~2e5 would overflow fp16)
    const float* a2;              // optional second source for k >= k_split (concat along K; same row addressing, k stride sk2)
    long long sk2;
    int k_split;                  // multiple of 8
    struct Row { long long off; int ok; };
    struct Raw { float v[8]; int k; };
    __device__ __forceinline__ Row row(int m) const {
        if (m >= M) return Row{0, 0};
        const int hi = m / m1, lo = m - hi * m1;
        return Row{hi * sm2 + lo * sm, 1};
    }
    __device__ __forceinline__ void issue(const Row& r, int k, Raw& o) const {
#pragma unroll
        for (int i = 0; i < 8; ++i) o.v[i] = 0.f;
        o.k = -1;
        if (!r.ok || k >= K) return;
        o.k = k;
        const bool second = a2 != nullptr && k >= k_split;
        const long long skk = second ? sk2 : sk;
        const float* p = second ? a2 + r.off + (long long)(k - k_split) * sk2 : a + r.off + (long long)k * sk;
        if (skk == 1 && k + 8 <= K && ((reinterpret_cast<size_t>(p) & 15) == 0)) {
            const float4 x = *reinterpret_cast<const float4*>(p), y = *reinterpret_cast<const float4*>(p + 4);
            o.v[0] = x.x; o.v[1] = x.y; o.v[2] = x.z; o.v[3] = x.w; o.v[4] = y.x; o.v[5] = y.y; o.v[6] = y.z; o.v[7] = y.w;
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (k + i < K) o.v[i] = p[(long long)i * skk];
        }
    }
    __device__ __forceinline__ void finish(const Raw& r, float (&v)[8]) const {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = r.v[i];
        if (kscale != nullptr && r.k >= 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (r.k + i < K) v[i] = v[i] * kscale[r.k + i] + kshift[r.k + i];
        }
    }
    __device__ __forceinline__ uint4 direct(const Raw&) const { return make_uint4(0, 0, 0, 0); }
};

// ---- A operand: fp32, strided, the common case without the loader's per-element predicates -------------------------------- //
// One source, no affine, K a multiple of 8, every element offset of one batch below 2^30 (checked at launch).  Rows beyond M are
// clamped to the last row instead of predicated (their products are never stored); every k-tile but a ragged last one is fetched
// as  uniform base (SGPR) + one 32-bit per-thread byte offset that does not change over the k loop  -- no address VALU, no
// exec-mask branches (the general loader spends ~250 VALU instructions and ~40 branches per k-tile and thread on them).
template <bool VEC>
struct ALFast {
    static constexpr bool kDirect = false, kUniformK = true;
    // k strided (VEC = false): in every use of this loader the ROWS are the contiguous index (pixels of an NCHW activation, latitudes of a
    // spectrum: a_sm == 1) -- stage rows-first (gemm.h: al_rows_first_t); harmless where they are not
    static constexpr bool kRowsFirst = !VEC;
    __device__ __forceinline__ bool rows_first() const { return true; }
    const float* a;
    int M, K, m1;
    long long sm, sm2, sk;
    struct Row { unsigned off; };
    struct Raw { float v[8]; };
    __device__ __forceinline__ Row row(int m) const {
        m = m < M ? m : M - 1;
        const int hi = m / m1, lo = m - hi * m1;
        return Row{(unsigned)(hi * sm2 + lo * sm)};
    }
    __device__ __forceinline__ void issue2(const Row& r, int k_tile, int k_slot, int bk, Raw& o) const {
        if (k_tile + bk <= K) {                                            // wave-uniform
            const float* base = a + (long long)k_tile * sk;                // uniform
            const unsigned vb = (r.off + (unsigned)k_slot * (unsigned)sk) * 4u;
            if constexpr (VEC) {
                const float4 x = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + vb);
                const float4 y = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base + 4) + vb);
                o.v[0] = x.x; o.v[1] = x.y; o.v[2] = x.z; o.v[3] = x.w; o.v[4] = y.x; o.v[5] = y.y; o.v[6] = y.z; o.v[7] = y.w;
            } else {
#pragma unroll
                for (int i = 0; i < 8; ++i) o.v[i] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base + (long long)i * sk) + vb);
            }
        } else {                                                           // ragged last k-tile: a chunk is inside K or outside (K % 8 == 0)
            const int k = k_tile + k_slot;
            const bool ok = k < K;
            const float* p = a + r.off + (long long)(ok ? k : 0) * sk;
#pragma unroll
            for (int i = 0; i < 8; ++i) { const float t = p[(long long)i * sk]; o.v[i] = ok ? t : 0.f; }
        }
    }
    __device__ __forceinline__ void issue(const Row&, int, Raw&) const {}
    __device__ __forceinline__ void finish(const Raw& r, float (&v)[8]) const {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = r.v[i];
    }
    __device__ __forceinline__ uint4 direct(const Raw&) const { return make_uint4(0, 0, 0, 0); }
};

__device__ __forceinline__ float swish(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }

// ---- epilogue: bias, residual before the activation, GELU, residual after, strided fp32 store ---- //
struct EpStrided {
    static constexpr bool kDualOrder = false;
    template <class TC> __device__ __forceinline__ void init(char*, int, int) const {}
    float* out;
    const float* bias;
    const float* res_pre;
    const float* res_post;
    int m1, act;
    long long sm, sm2, sn;
    __device__ __forceinline__ long long addr(int m, int n) const {
        const int hi = m / m1, lo = m - hi * m1;
        return hi * sm2 + lo * sm + (long long)n * sn;
    }
    __device__ __forceinline__ float post(float v, long long o, int n) const {
        if (bias) v += bias[n];
        if (res_pre) v += res_pre[o];
        if (act == 1) v = gelu_erf(v);
        else if (act == 2) v = swish(v);
        if (res_post) v += res_post[o];
        return v;
    }
    // Loads before stores (epilogues.h): on gfx950 a load issued after a store waits for that store's acknowledgement, so the
    // bias is read once up front and the residual operands of row group a + 1 are in flight while group a is computed and
    // stored; only the rare element-wise path (two-level rows, tails, unaligned) loads inside the store loop.
    template <class TC, bool SWAP>
    __device__ __forceinline__ void run(f32x4 (&acc)[TC::FM][TC::FN], int m0w, int n0w, int lane, int, int, char*, int M, int N, int) const {
        constexpr int FM = TC::FM, FN = TC::FN;
        const int l15 = lane & 15, l4 = (lane >> 4) * 4;
        const bool aligned = (reinterpret_cast<size_t>(out) & 15) == 0 && (res_pre == nullptr || (reinterpret_cast<size_t>(res_pre) & 15) == 0) &&
                             (res_post == nullptr || (reinterpret_cast<size_t>(res_post) & 15) == 0);
        float4 bv[FN];                                   // SWAP: bias of 4 consecutive columns; else .x = the lane's column
#pragma unroll
        for (int b = 0; b < FN; ++b) {
            const int n = n0w + b * 16 + (SWAP ? l4 : l15);
            bv[b] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (bias != nullptr && n < N) {
                if (SWAP) {
                    bv[b].x = bias[n];
                    if (n + 1 < N) bv[b].y = bias[n + 1];
                    if (n + 2 < N) bv[b].z = bias[n + 2];
                    if (n + 3 < N) bv[b].w = bias[n + 3];
                } else {
                    bv[b].x = bv[b].y = bv[b].z = bv[b].w = bias[n];
                }
            }
        }
        long long off[2][FN];
        bool vec[2][FN];
        float4 rp[2][FN], rq[2][FN];
        auto prefetch = [&](int a, int s) {
#pragma unroll
            for (int b = 0; b < FN; ++b) {
                const int m = m0w + a * 16 + (SWAP ? l15 : l4), n = n0w + b * 16 + (SWAP ? l4 : l15);
                vec[s][b] = false;
                rp[s][b] = make_float4(0.f, 0.f, 0.f, 0.f);
                rq[s][b] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (m >= M || n >= N) continue;
                off[s][b] = addr(m, n);
                const bool v = (SWAP ? (sn == 1 && n + 3 < N) : (sm == 1 && m1 >= M && m + 3 < M)) && (off[s][b] & 3) == 0 && aligned;
                vec[s][b] = v;
                if (v) {
                    if (res_pre) rp[s][b] = *reinterpret_cast<const float4*>(res_pre + off[s][b]);
                    if (res_post) rq[s][b] = *reinterpret_cast<const float4*>(res_post + off[s][b]);
                }
            }
        };
        prefetch(0, 0);
#pragma unroll
        for (int a = 0; a < FM; ++a) {
            const int s = a & 1;
            if (a + 1 < FM) prefetch(a + 1, s ^ 1);
#pragma unroll
            for (int b = 0; b < FN; ++b) {
                const int m = m0w + a * 16 + (SWAP ? l15 : l4), n = n0w + b * 16 + (SWAP ? l4 : l15);
                if (m >= M || n >= N) continue;
                if (vec[s][b]) {
                    float v[4] = {acc[a][b][0] + bv[b].x + rp[s][b].x, acc[a][b][1] + bv[b].y + rp[s][b].y,
                                  acc[a][b][2] + bv[b].z + rp[s][b].z, acc[a][b][3] + bv[b].w + rp[s][b].w};
                    if (act == 1) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
                    } else if (act == 2) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = swish(v[r]);
                    }
                    *reinterpret_cast<float4*>(out + off[s][b]) = make_float4(v[0] + rq[s][b].x, v[1] + rq[s][b].y, v[2] + rq[s][b].z, v[3] + rq[s][b].w);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int mm = SWAP ? m : m + r, nn = SWAP ? n + r : n;
                        if (mm < M && nn < N) {
                            const long long o = addr(mm, nn);
                            out[o] = post(acc[a][b][r], o, nn);
                        }
                    }
                }
            }
        }
    }
};

typedef TileCfg<128, 256, 32, 2, 4> TG;      // wide N: the fp32 A operand is fetched and split once per 256 output columns

struct BatchStrides { long long a, w, o; int k_lo_step, m_cap0, m_cap_step, xcd_remap; };

}  // namespace skp
