// A-operand loaders for the register-staged GEMM (gemm.h): the two stages whose A operand is computed on the fly
// (im2col of the raw fp32 state; 2x2 merge + LayerNorm of the residual planes).  A loader turns "row m, columns
// k..k+7" of the logical A matrix into 8 fp32 values that the GEMM rounds / splits on their way into LDS.
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
template <class T>
struct ALMergeLN {
    static constexpr bool kDirect = false;
    const T* x;              // residual-stream planes [Z*H1*W1][C], blocked layout; lo plane at + plane
    long long plane;
    const float2* stats;     // [M] (mean, rstd)
    const float* gamma;      // [4C]
    const float* beta;
    int H1, W1, H2, W2, C, M;
    struct Row { long long tok00; int has_h1; float mean, rstd; };
    struct Raw { uint4 hi, lo; float mean, rstd; int k; };
    __device__ __forceinline__ Row row(int m) const {
        if (m >= M) return Row{-1, 0, 0.f, 0.f};
        const int hw = H2 * W2;
        const int z = m / hw, rem = m - z * hw;
        const int h = rem / W2, w = rem - h * W2;
        const float2 s = stats[m];
        return Row{((long long)z * H1 + 2 * h) * W1 + 2 * w, (2 * h + 1 < H1) ? 1 : 0, s.x, s.y};
    }
    __device__ __forceinline__ void issue(const Row& r, int k, Raw& o) const {
        o.hi = make_uint4(0, 0, 0, 0); o.lo = make_uint4(0, 0, 0, 0); o.k = -1; o.mean = r.mean; o.rstd = r.rstd;
        if (r.tok00 < 0 || k >= 4 * C) return;
        const int q = k / C, c = k - q * C, dh = q >> 1, dw = q & 1;
        o.k = k;
        if (dh == 0 || r.has_h1) {
            const T* p = x + blk_off(r.tok00 + (long long)dh * W1 + dw, c, C);
            o.hi = *reinterpret_cast<const uint4*>(p);
            o.lo = *reinterpret_cast<const uint4*>(p + plane);
        }
    }
    __device__ __forceinline__ void finish(const Raw& r, float (&v)[8]) const {
        if (r.k < 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = 0.f;
            return;
        }
        const typename OpT<T>::v8 h = as_v8<T>(r.hi), l = as_v8<T>(r.lo);
        const float4 g0 = *reinterpret_cast<const float4*>(gamma + r.k), g1 = *reinterpret_cast<const float4*>(gamma + r.k + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(beta + r.k), b1 = *reinterpret_cast<const float4*>(beta + r.k + 4);
        const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        const float b[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = (((float)h[i] + (float)l[i]) - r.mean) * r.rstd * g[i] + b[i];
    }
    __device__ __forceinline__ uint4 direct(const Raw&) const { return make_uint4(0, 0, 0, 0); }
};

}  // namespace skp
