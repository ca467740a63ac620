// A-operand loaders for gemm.h.  A loader turns "row m, columns k..k+7" of the logical A matrix
// into 8 fp32 values (or, for 16-bit sources, directly into one 16-byte LDS chunk).
//
//   Row  row(m)                 per-row context, computed once per block row (index math lives here)
//   void issue(Row, k, Raw&)    issue the global loads of one 8-element chunk (must zero-fill k >= K)
//   void finish(Raw, v[8])      turn the raw data into the fp32 values the GEMM rounds/splits
#pragma once
#include "common.h"

namespace skp {

__device__ __forceinline__ float4 f4zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void f4x2_to_arr(const float4& a, const float4& b, float (&v)[8]) {
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}

// ---- plain / indexed rows of an fp32 matrix ------------------------------- //
// idx (optional): logical row m reads source row idx[m]; idx[m] < 0 is a zero (padding) row.
// Used for the residual stream (fc1), and with idx = window table for the QKV window gather
// (roll + pad + window partition of EarthSpecificBlock).
struct ALRowsF32 {
    static constexpr bool kDirect = false;
    const float* x;
    const int* idx;
    int ld, M, K, row_off;
    struct Row { const float* p; };
    struct Raw { float4 a, b; };
    __device__ __forceinline__ Row row(int m) const {
        if (m >= M) return Row{nullptr};
        const int s = idx ? idx[m] : m + row_off;
        return Row{s < 0 ? nullptr : x + (long long)s * ld};
    }
    __device__ __forceinline__ void issue(const Row& r, int k, Raw& o) const {
        if (r.p != nullptr && k < K) {
            o.a = *reinterpret_cast<const float4*>(r.p + k);
            o.b = *reinterpret_cast<const float4*>(r.p + k + 4);
        } else {
            o.a = f4zero(); o.b = f4zero();
        }
    }
    __device__ __forceinline__ void finish(const Raw& r, float (&v)[8]) const { f4x2_to_arr(r.a, r.b, v); }
    __device__ __forceinline__ uint4 direct(const Raw&) const { return make_uint4(0, 0, 0, 0); }
};

// ---- plain rows of a 16-bit matrix (single-term modes only) ---------------- //
template <class T>
struct ALRowsT {
    static constexpr bool kDirect = true;
    const T* x;
    int ld, M, K;
    struct Row { const T* p; };
    typedef uint4 Raw;
    __device__ __forceinline__ Row row(int m) const { return Row{m < M ? x + (long long)m * ld : nullptr}; }
    __device__ __forceinline__ void issue(const Row& r, int k, Raw& o) const {
        o = (r.p != nullptr && k < K) ? *reinterpret_cast<const uint4*>(r.p + k) : make_uint4(0, 0, 0, 0);
    }
    __device__ __forceinline__ void finish(const Raw&, float (&)[8]) const {}
    __device__ __forceinline__ uint4 direct(const Raw& r) const { return r; }
};

// activation rows stored as fp32 (split modes) or T (single-term modes)
template <class P, class S> struct ALRowsAct;
template <class P> struct ALRowsAct<P, float> { typedef ALRowsF32 type;
    static __device__ __host__ type make(const float* x, int ld, int M, int K) { return type{x, nullptr, ld, M, K, 0}; } };
template <class P> struct ALRowsAct<P, typename P::T> { typedef ALRowsT<typename P::T> type;
    static __device__ __host__ type make(const typename P::T* x, int ld, int M, int K) { return type{x, ld, M, K}; } };

// ---- concat(skip, x) rows for PatchRecovery -------------------------------- //
struct ALConcat2 {
    static constexpr bool kDirect = false;
    const float* a;
    const float* b;
    int C1, M, row_off;      // both sources are [rows][C1]
    struct Row { const float* pa; const float* pb; };
    struct Raw { float4 a, b; };
    __device__ __forceinline__ Row row(int m) const {
        if (m >= M) return Row{nullptr, nullptr};
        const long long s = (long long)(m + row_off) * C1;
        return Row{a + s, b + s};
    }
    __device__ __forceinline__ void issue(const Row& r, int k, Raw& o) const {
        if (r.pa != nullptr && k < 2 * C1) {
            const float* p = k < C1 ? r.pa + k : r.pb + (k - C1);
            o.a = *reinterpret_cast<const float4*>(p);
            o.b = *reinterpret_cast<const float4*>(p + 4);
        } else {
            o.a = f4zero(); o.b = f4zero();
        }
    }
    __device__ __forceinline__ void finish(const Raw& r, float (&v)[8]) const { f4x2_to_arr(r.a, r.b, v); }
    __device__ __forceinline__ uint4 direct(const Raw&) const { return make_uint4(0, 0, 0, 0); }
};

// ---- im2col of the raw state for PatchEmbedding ---------------------------- //
// Upper air: token (zt, h, w), zt in 0..6; k = ((v*2+dz)*4+dh)*4+dw, K = 160 (Conv3d weight order).
// Surface:   token (h, w);                k = (c*4+dh)*4+dw,         K = 112, c = 4 state vars + 3 masks.
// Values are normalised (x - mean_c) * istd_c; padded latitudes / the 14th level are zero AFTER
// normalisation.  One chunk = two latitude rows x four longitudes = two float4 loads.
struct Im2colRaw { float4 a, b; float mean, istd; int va, vb; };
__device__ __forceinline__ void im2col_finish(const Im2colRaw& r, float (&v)[8]) {
    v[0] = r.va ? (r.a.x - r.mean) * r.istd : 0.f; v[1] = r.va ? (r.a.y - r.mean) * r.istd : 0.f;
    v[2] = r.va ? (r.a.z - r.mean) * r.istd : 0.f; v[3] = r.va ? (r.a.w - r.mean) * r.istd : 0.f;
    v[4] = r.vb ? (r.b.x - r.mean) * r.istd : 0.f; v[5] = r.vb ? (r.b.y - r.mean) * r.istd : 0.f;
    v[6] = r.vb ? (r.b.z - r.mean) * r.istd : 0.f; v[7] = r.vb ? (r.b.w - r.mean) * r.istd : 0.f;
}

struct ALIm2colUpper {
    static constexpr bool kDirect = false;
    const float* state;      // [69][n_lat][n_lon]
    const float* mean;       // [69]
    const float* istd;       // [69]
    int n_lat, n_lon, lat_top, H1, W1, n_levels, M;
    struct Row { int zt, lat0, col; };
    typedef Im2colRaw Raw;
    __device__ __forceinline__ Row row(int m) const {
        if (m >= M) return Row{-1, 0, 0};
        const int hw = H1 * W1;
        const int zt = m / hw, rem = m - zt * hw;
        const int h = rem / W1, w = rem - h * W1;
        return Row{zt, 4 * h - lat_top, 4 * w};
    }
    __device__ __forceinline__ void issue(const Row& r, int k, Raw& o) const {
        const int kc = k >> 3;
        o.a = f4zero(); o.b = f4zero(); o.mean = 0.f; o.istd = 0.f; o.va = 0; o.vb = 0;
        if (r.zt < 0 || kc >= 20) return;
        const int v = kc >> 2, dz = (kc >> 1) & 1, dhp = kc & 1;
        const int level = 2 * r.zt + dz;
        if (level >= n_levels) return;
        const int ch = v * n_levels + level;
        const int la = r.lat0 + 2 * dhp, lb = la + 1;
        const float* base = state + (long long)ch * n_lat * n_lon + r.col;
        o.mean = mean[ch]; o.istd = istd[ch];
        if (la >= 0 && la < n_lat) { o.a = *reinterpret_cast<const float4*>(base + (long long)la * n_lon); o.va = 1; }
        if (lb >= 0 && lb < n_lat) { o.b = *reinterpret_cast<const float4*>(base + (long long)lb * n_lon); o.vb = 1; }
    }
    __device__ __forceinline__ void finish(const Raw& r, float (&v)[8]) const { im2col_finish(r, v); }
    __device__ __forceinline__ uint4 direct(const Raw&) const { return make_uint4(0, 0, 0, 0); }
};

struct ALIm2colSurface {
    static constexpr bool kDirect = false;
    const float* state;      // [69][n_lat][n_lon]; surface vars are channels surf0..surf0+3
    const float* masks;      // [3][n_lat][n_lon]
    const float* mean;
    const float* istd;
    int n_lat, n_lon, lat_top, H1, W1, surf0, M;
    struct Row { int ok, lat0, col; };
    typedef Im2colRaw Raw;
    __device__ __forceinline__ Row row(int m) const {
        if (m >= M) return Row{0, 0, 0};
        const int h = m / W1, w = m - h * W1;
        return Row{1, 4 * h - lat_top, 4 * w};
    }
    __device__ __forceinline__ void issue(const Row& r, int k, Raw& o) const {
        const int kc = k >> 3;
        o.a = f4zero(); o.b = f4zero(); o.mean = 0.f; o.istd = 1.f; o.va = 0; o.vb = 0;
        if (!r.ok || kc >= 14) return;
        const int c = kc >> 1, dhp = kc & 1;
        const float* base;
        if (c < 4) { base = state + (long long)(surf0 + c) * n_lat * n_lon; o.mean = mean[surf0 + c]; o.istd = istd[surf0 + c]; }
        else       { base = masks + (long long)(c - 4) * n_lat * n_lon; }
        base += r.col;
        const int la = r.lat0 + 2 * dhp, lb = la + 1;
        if (la >= 0 && la < n_lat) { o.a = *reinterpret_cast<const float4*>(base + (long long)la * n_lon); o.va = 1; }
        if (lb >= 0 && lb < n_lat) { o.b = *reinterpret_cast<const float4*>(base + (long long)lb * n_lon); o.vb = 1; }
    }
    __device__ __forceinline__ void finish(const Raw& r, float (&v)[8]) const { im2col_finish(r, v); }
    __device__ __forceinline__ uint4 direct(const Raw&) const { return make_uint4(0, 0, 0, 0); }
};

// ---- DownSample: 2x2 merge + LayerNorm(4C) as the GEMM prologue ------------- //
// merged row (z, h', w') ; k = (dh*2+dw)*C + c reads token (z, 2h'+dh, 2w'+dw) (zero if 2h'+dh >= H1),
// normalised with the per-row statistics computed by merge_stats_kernel (stats.hip).
struct ALMergeLN {
    static constexpr bool kDirect = false;
    const float* x;          // [Z*H1*W1][C]
    const float2* stats;     // [M] (mean, rstd)
    const float* gamma;      // [4C]
    const float* beta;
    int H1, W1, H2, W2, C, M;
    struct Row { long long tok00; int has_h1; float mean, rstd; };
    struct Raw { float4 a, b; int k; };
    __device__ __forceinline__ Row row(int m) const {
        if (m >= M) return Row{-1, 0, 0.f, 0.f};
        const int hw = H2 * W2;
        const int z = m / hw, rem = m - z * hw;
        const int h = rem / W2, w = rem - h * W2;
        const float2 s = stats[m];
        return Row{((long long)z * H1 + 2 * h) * W1 + 2 * w, (2 * h + 1 < H1) ? 1 : 0, s.x, s.y};
    }
    __device__ __forceinline__ void issue(const Row& r, int k, Raw& o) const {
        o.a = f4zero(); o.b = f4zero(); o.k = -1;
        if (r.tok00 < 0 || k >= 4 * C) return;
        const int q = k / C, c = k - q * C, dh = q >> 1, dw = q & 1;
        // the row's statistics travel in the Raw via mean/rstd folded at finish(); keep k for gamma/beta
        o.k = k;
        if (dh == 0 || r.has_h1) {
            const float* p = x + (r.tok00 + (long long)dh * W1 + dw) * C + c;
            o.a = *reinterpret_cast<const float4*>(p);
            o.b = *reinterpret_cast<const float4*>(p + 4);
        }
        o.a.x = (o.a.x - r.mean) * r.rstd; o.a.y = (o.a.y - r.mean) * r.rstd;
        o.a.z = (o.a.z - r.mean) * r.rstd; o.a.w = (o.a.w - r.mean) * r.rstd;
        o.b.x = (o.b.x - r.mean) * r.rstd; o.b.y = (o.b.y - r.mean) * r.rstd;
        o.b.z = (o.b.z - r.mean) * r.rstd; o.b.w = (o.b.w - r.mean) * r.rstd;
    }
    __device__ __forceinline__ void finish(const Raw& r, float (&v)[8]) const {
        if (r.k < 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = 0.f;
            return;
        }
        const float4 g0 = *reinterpret_cast<const float4*>(gamma + r.k), g1 = *reinterpret_cast<const float4*>(gamma + r.k + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(beta + r.k), b1 = *reinterpret_cast<const float4*>(beta + r.k + 4);
        v[0] = r.a.x * g0.x + b0.x; v[1] = r.a.y * g0.y + b0.y; v[2] = r.a.z * g0.z + b0.z; v[3] = r.a.w * g0.w + b0.w;
        v[4] = r.b.x * g1.x + b1.x; v[5] = r.b.y * g1.y + b1.y; v[6] = r.b.z * g1.z + b1.z; v[7] = r.b.w * g1.w + b1.w;
    }
    __device__ __forceinline__ uint4 direct(const Raw&) const { return make_uint4(0, 0, 0, 0); }
};

}  // namespace skp
