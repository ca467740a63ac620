// GraphCast building blocks behind include/skyrim_graphcast.h (interaction-network MLPs on the icosahedral multi-mesh).
//
// skgc_gather_gemm: the first Linear of every MLP.  Its input row is never materialised: the loader assembles
// concat(edge latent, sender latent, receiver latent) from the three row-major sources through the edge's index arrays
// (k-contiguous 32-byte reads), so a 3.1 M-edge x 1536-wide matrix (19 GB) stays virtual.  Same 3-term fp16 MFMA pipeline as
// the SFNO GEMMs (strided_gemm.h / gemm.h); LayerNorm and the receiver sum are small HBM-bound kernels.
#include <cstdlib>
#include <type_traits>
#include "../../include/skyrim_graphcast.h"
#include "strided_gemm.h"

namespace skp {

struct ALGather {
    static constexpr bool kDirect = false;
    const float* src[3];
    const int* idx[3];
    long long ld[3];
    int k0[4];                    // segment s covers k0[s] <= k < k0[s + 1]
    int n_src, M, K;
    const float* kscale;
    const float* kshift;
    struct Row { long long r[3]; int ok; };
    struct Raw { float v[8]; int k; };
    __device__ __forceinline__ Row row(int m) const {
        Row o;
        o.ok = m < M;
#pragma unroll
        for (int s = 0; s < 3; ++s) o.r[s] = (o.ok && s < n_src) ? (long long)(idx[s] ? idx[s][m] : m) * ld[s] : 0;
        return o;
    }
    __device__ __forceinline__ void issue(const Row& r, int k, Raw& o) const {
#pragma unroll
        for (int i = 0; i < 8; ++i) o.v[i] = 0.f;
        o.k = -1;
        if (!r.ok || k >= K) return;
        o.k = k;
        const int s = k >= k0[2] ? 2 : (k >= k0[1] ? 1 : 0);            // chunks never straddle segments (widths % 8 == 0)
        const float* p = src[s] + r.r[s] + (k - k0[s]);
        if (k + 8 <= k0[s + 1] && (reinterpret_cast<size_t>(p) & 15) == 0) {
            const float4 x = *reinterpret_cast<const float4*>(p), y = *reinterpret_cast<const float4*>(p + 4);
            o.v[0] = x.x; o.v[1] = x.y; o.v[2] = x.z; o.v[3] = x.w; o.v[4] = y.x; o.v[5] = y.y; o.v[6] = y.z; o.v[7] = y.w;
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (k + i < k0[s + 1]) o.v[i] = p[i];
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

// A operand = swish( sum_s src[s][ idx[s] ? idx[s][m] : m ][k] ): the first Linear of an edge MLP by distributivity.
//   fc1(concat(e, v_s[send], v_r[recv])) = e W_e^T + (v_s W_s^T)[send] + (v_r W_r^T)[recv] + b
// The two node terms are computed once per NODE (8 edges share a mesh node, 3 a grid node), the edge term once per layer (or once
// per model where the edge latent is input-independent), and this loader adds the gathered rows up on the fly and applies the
// activation -- the hidden activation of the edge MLP is never materialised (12.8 GB of traffic per step for the mesh->grid edges),
// and 44 % of the step's FLOPs are gone.
struct ALSumGather {
    static constexpr bool kDirect = false;
    const float* src[3];
    const int* idx[3];
    long long ld[3];
    int n_src, M, K, act;
    struct Row { long long r[3]; int ok; };
    struct Raw { float4 lo[3], hi[3]; int ok; };
    __device__ __forceinline__ Row row(int m) const {
        Row o;
        o.ok = m < M;
#pragma unroll
        for (int s = 0; s < 3; ++s) o.r[s] = (o.ok && s < n_src) ? (long long)(idx[s] ? idx[s][m] : m) * ld[s] : 0;
        return o;
    }
    __device__ __forceinline__ void issue(const Row& r, int k, Raw& o) const {
        o.ok = r.ok && k < K;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            o.lo[s] = make_float4(0.f, 0.f, 0.f, 0.f); o.hi[s] = o.lo[s];
            if (o.ok && s < n_src) {                       // K % 8 == 0 and 16-byte aligned rows (checked by the launcher)
                const float* p = src[s] + r.r[s] + k;
                o.lo[s] = *reinterpret_cast<const float4*>(p);
                o.hi[s] = *reinterpret_cast<const float4*>(p + 4);
            }
        }
    }
    __device__ __forceinline__ void finish(const Raw& r, float (&v)[8]) const {
        const float4 a = r.lo[0], b = r.lo[1], c = r.lo[2], d = r.hi[0], e = r.hi[1], f = r.hi[2];
        v[0] = (a.x + b.x) + c.x; v[1] = (a.y + b.y) + c.y; v[2] = (a.z + b.z) + c.z; v[3] = (a.w + b.w) + c.w;
        v[4] = (d.x + e.x) + f.x; v[5] = (d.y + e.y) + f.y; v[6] = (d.z + e.z) + f.z; v[7] = (d.w + e.w) + f.w;
        if (act == 2 && r.ok) {
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = swish(v[i]);
        }
    }
    __device__ __forceinline__ uint4 direct(const Raw&) const { return make_uint4(0, 0, 0, 0); }
};

// second Linear of an MLP fused with its LayerNorm (+ residual): the tile spans the whole latent (N = 512 = LayerNorm width), so
// the pre-norm activations never go to HBM (for the 3.1 M mesh->grid edges that is 12.8 GB of traffic per step).  The weight is
// prepared in perm8 row order (common.h): a lane's accumulators of a fragment pair are 8 consecutive columns -> 32-byte row pieces.
template <bool RES>
struct SinkRowsF32 {
    float* out;
    const float* res;            // RES: out[row] = res[row] + y (out may alias res)
    static constexpr bool kLoads = RES;
    __device__ __forceinline__ f32x8 load(long long row, int ld, int c) const {
        f32x8 o;
        o.a = *reinterpret_cast<const float4*>(res + row * ld + c);
        o.b = *reinterpret_cast<const float4*>(res + row * ld + c + 4);
        return o;
    }
    __device__ __forceinline__ void put(long long row, int ld, int c, const float (&y)[8], const f32x8& old) const {
        float* p = out + row * ld + c;
        if constexpr (RES) {
            *reinterpret_cast<float4*>(p) = make_float4(old.a.x + y[0], old.a.y + y[1], old.a.z + y[2], old.a.w + y[3]);
            *reinterpret_cast<float4*>(p + 4) = make_float4(old.b.x + y[4], old.b.y + y[5], old.b.z + y[6], old.b.w + y[7]);
        } else {
            *reinterpret_cast<float4*>(p) = make_float4(y[0], y[1], y[2], y[3]);
            *reinterpret_cast<float4*>(p + 4) = make_float4(y[4], y[5], y[6], y[7]);
        }
    }
};

// the tile spans the latent (N = 512): 128 rows x 512 columns, 8 waves as 2 x 4 with 64 x 128 wave tiles (96 MFMAs per wave and
// k-step against one 64 KB weight k-tile; the 64-row tile TLN64 re-stages that weight tile twice as often)
typedef TileCfg<64, 512, 32, 1, 8> TLN64;
typedef TileCfg<128, 512, 32, 2, 4> TLN128;
// groups of three rows summed after the LayerNorm (skgc_sum_desc::group == 3): 2 x 4 waves with 48 x 128 wave tiles, fragment a of a
// wave tile = member a of its 16 groups, so the sum over the members is a sum over a lane's own registers
typedef TileCfg<96, 512, 32, 2, 4> TLN96;
static int ln_tile_rows() { static const int v = [] { const char* e = getenv("SKGC_LN_TILE"); return e ? atoi(e) : 128; }(); return v; }

template <class TLN, bool RES>
__global__ void __launch_bounds__(TLN::THREADS) linear_ln_kernel(const GemmArgs<PrecF16x3, ALStrided, EpLayerNorm<RowMapIndexed, SinkRowsF32<RES>>> g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gemm_body<PrecF16x3, TLN, ALStrided, EpLayerNorm<RowMapIndexed, SinkRowsF32<RES>>, true>(g, smem);
}

template <class TLN, bool RES>
__global__ void __launch_bounds__(TLN::THREADS) sum_linear_ln_kernel(const GemmArgs<PrecF16x3, ALSumGather, EpLayerNorm<RowMapIndexed, SinkRowsF32<RES>>> g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gemm_body<PrecF16x3, TLN, ALSumGather, EpLayerNorm<RowMapIndexed, SinkRowsF32<RES>>, true>(g, smem);
}

__global__ void __launch_bounds__(TLN96::THREADS) sum3_linear_ln_kernel(const GemmArgs<PrecF16x3, ALSumGather, EpLayerNorm<RowMapIndexed, SinkRowsF32<false>, 3>> g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gemm_body<PrecF16x3, TLN96, ALSumGather, EpLayerNorm<RowMapIndexed, SinkRowsF32<false>, 3>, true>(g, smem);
}

__global__ void __launch_bounds__(TG::THREADS) gather_gemm_kernel(const GemmArgs<PrecF16x3, ALGather, EpStrided> g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gemm_body<PrecF16x3, TG, ALGather, EpStrided, true>(g, smem);
}

// one wave per row; N <= 1024
__global__ void __launch_bounds__(256) layer_norm_kernel(const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const float* res, float* out, long long rows, int N) {
    const int lane = threadIdx.x & 63;
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float* xr = x + r * N;
    float v[16];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = lane + i * 64;
        v[i] = c < N ? xr[c] : 0.f;
        s += v[i];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)N;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const float d = (lane + i * 64 < N) ? v[i] - mean : 0.f;
        q += d * d;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = rsqrtf(q / (float)N + 1e-5f);
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = lane + i * 64;
        if (c < N) {
            float y = (v[i] - mean) * rstd * gamma[c] + beta[c];
            if (res) y += res[r * N + c];
            out[r * N + c] = y;
        }
    }
}

// one wave per node, float4 per lane per pass (N % 4 == 0)
// acc (optional): the residual update of the edge latents rides along, acc[j] += e[j], since every edge row is read here anyway
__global__ void __launch_bounds__(256) segment_sum_kernel(const float* __restrict__ e, const int* __restrict__ offsets, float* __restrict__ out,
                                                          float* acc, int n_nodes, int N) {
    const int lane = threadIdx.x & 63;
    const int v = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (v >= n_nodes) return;
    const int j0 = offsets[v], j1 = offsets[v + 1];
    for (int c = lane * 4; c < N; c += 256) {
        float4 sum = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int j = j0; j < j1; ++j) {
            const float4 t = *reinterpret_cast<const float4*>(e + (long long)j * N + c);
            sum.x += t.x; sum.y += t.y; sum.z += t.z; sum.w += t.w;
            if (acc != nullptr) {
                float4 u = *reinterpret_cast<float4*>(acc + (long long)j * N + c);
                u.x += t.x; u.y += t.y; u.z += t.z; u.w += t.w;
                *reinterpret_cast<float4*>(acc + (long long)j * N + c) = u;
            }
        }
        *reinterpret_cast<float4*>(out + (long long)v * N + c) = sum;
    }
}

}  // namespace skp

using namespace skp;

template <class TLN>
static int linear_layer_norm_t(const float* a, long long lda, int K, const void* w, long long w_plane, int ldw, const float* bias, const float* gamma,
                               const float* beta, const float* res, float* out, long long rows, void* stream) {
    constexpr int N = TLN::BN;
    const dim3 grid(1, (unsigned)((rows + TLN::BM - 1) / TLN::BM));
    if (grid.y > 65535 * 16) return SKGC_E_ARG;
    constexpr int smem = gemm_smem_bytes<PrecF16x3, TLN>() + kEpiScratch;
    const ALStrided al{a, (int)rows, K, 1 << 30, lda, 0, 1, nullptr, nullptr, nullptr, 0, 0};
    hipStream_t st = static_cast<hipStream_t>(stream);
    auto launch = [&](auto res_c) {
        constexpr bool RES = decltype(res_c)::value;
        typedef EpLayerNorm<RowMapIndexed, SinkRowsF32<RES>> EP;
        GemmArgs<PrecF16x3, ALStrided, EP> g;
        g.al = al;
        g.ep = EP{RowMapIndexed{nullptr}, SinkRowsF32<RES>{out, res}, bias, gamma, beta, 1e-5f};
        g.W = static_cast<const f16*>(w);
        g.w_plane = w_plane;
        g.ldw = ldw;
        g.M = (int)rows; g.N = N; g.K = K;
        auto kern = linear_ln_kernel<TLN, RES>;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, grid, dim3(TLN::THREADS), smem, st, g);
        return hipGetLastError();
    };
    const hipError_t e = res ? launch(std::true_type{}) : launch(std::false_type{});
    return e == hipSuccess ? 0 : SKGC_E_HIP;
}

template <class TLN>
static int sum_linear_layer_norm_t(const skgc_sum_desc* d, const ALSumGather& al, void* stream) {
    constexpr int N = TLN::BN;
    const dim3 grid(1, (unsigned)((d->rows + TLN::BM - 1) / TLN::BM));
    constexpr int smem = gemm_smem_bytes<PrecF16x3, TLN>() + kEpiScratch;
    hipStream_t st = static_cast<hipStream_t>(stream);
    auto launch = [&](auto res_c) {
        constexpr bool RES = decltype(res_c)::value;
        typedef EpLayerNorm<RowMapIndexed, SinkRowsF32<RES>> EP;
        GemmArgs<PrecF16x3, ALSumGather, EP> g;
        g.al = al;
        g.ep = EP{RowMapIndexed{nullptr}, SinkRowsF32<RES>{d->out, d->res}, d->bias, d->gamma, d->beta, 1e-5f};
        g.W = static_cast<const f16*>(d->w);
        g.w_plane = d->w_plane;
        g.ldw = d->ldw;
        g.M = (int)d->rows; g.N = N; g.K = d->K;
        auto kern = sum_linear_ln_kernel<TLN, RES>;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, grid, dim3(TLN::THREADS), smem, st, g);
        return hipGetLastError();
    };
    const hipError_t e = d->res ? launch(std::true_type{}) : launch(std::false_type{});
    return e == hipSuccess ? 0 : SKGC_E_HIP;
}

// rows = 48 x ceil(groups / 16) virtual rows: row 48 t + 16 a + l is member a of group 16 t + l (the index arrays carry that order)
static int sum3_linear_layer_norm(const skgc_sum_desc* d, const ALSumGather& al, void* stream) {
    typedef EpLayerNorm<RowMapIndexed, SinkRowsF32<false>, 3> EP;
    const long long vrows = (d->rows + 15) / 16 * 48;
    if (vrows > 0x7fffffff) return SKGC_E_ARG;
    GemmArgs<PrecF16x3, ALSumGather, EP> g;
    g.al = al;
    g.al.M = (int)vrows;
    g.ep = EP{RowMapIndexed{nullptr}, SinkRowsF32<false>{d->out, nullptr}, d->bias, d->gamma, d->beta, 1e-5f, (int)d->rows};
    g.W = static_cast<const f16*>(d->w);
    g.w_plane = d->w_plane;
    g.ldw = d->ldw;
    g.M = (int)vrows; g.N = TLN96::BN; g.K = d->K;
    constexpr int smem = gemm_smem_bytes<PrecF16x3, TLN96>() + kEpiScratch;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(sum3_linear_ln_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return SKGC_E_HIP;
    hipLaunchKernelGGL(sum3_linear_ln_kernel, dim3(1, (unsigned)(vrows / TLN96::BM + (vrows % TLN96::BM != 0))), dim3(TLN96::THREADS), smem, static_cast<hipStream_t>(stream), g);
    return hipGetLastError() == hipSuccess ? 0 : SKGC_E_HIP;
}

extern "C" {

int skgc_abi_version(void) { return SKGC_ABI_VERSION; }

int skgc_gather_gemm(const skgc_gather_gemm_desc* d, void* stream) {
    if (!d || !d->w || !d->out || d->M <= 0 || d->N <= 0 || d->n_src < 1 || d->n_src > 3 || (d->ldw & 7) || d->ldo < d->N ||
        (d->act != 0 && d->act != 2) || (d->kscale == nullptr) != (d->kshift == nullptr))
        return SKGC_E_ARG;
    GemmArgs<PrecF16x3, ALGather, EpStrided> g;
    int k = 0;
    for (int s = 0; s < 3; ++s) {
        g.al.k0[s] = k;
        g.al.src[s] = nullptr; g.al.idx[s] = nullptr; g.al.ld[s] = 0;
        if (s < d->n_src) {
            if (!d->src[s] || d->width[s] <= 0 || d->ld[s] < d->width[s] || (s + 1 < d->n_src && (d->width[s] & 7))) return SKGC_E_ARG;
            g.al.src[s] = d->src[s]; g.al.idx[s] = d->idx[s]; g.al.ld[s] = d->ld[s];
            k += d->width[s];
        }
    }
    g.al.k0[3] = k;
    if (d->ldw < k) return SKGC_E_ARG;
    g.al.n_src = d->n_src; g.al.M = d->M; g.al.K = k; g.al.kscale = d->kscale; g.al.kshift = d->kshift;
    g.ep = EpStrided{d->out, d->bias, nullptr, nullptr, 1 << 30, d->act, d->ldo, 0, 1};
    g.W = static_cast<const f16*>(d->w);
    g.w_plane = d->w_plane;
    g.ldw = d->ldw;
    g.M = d->M; g.N = d->N; g.K = k;
    const dim3 grid((d->N + TG::BN - 1) / TG::BN, (d->M + TG::BM - 1) / TG::BM);
    if (grid.y > 65535) {
        // more than 8.3 M rows: not needed (3.1 M edges), refuse rather than wrap
        return SKGC_E_ARG;
    }
    constexpr int smem = gemm_smem_bytes<PrecF16x3, TG>() + kEpiScratch;
    hipLaunchKernelGGL(gather_gemm_kernel, grid, dim3(TG::THREADS), smem, static_cast<hipStream_t>(stream), g);
    return hipGetLastError() == hipSuccess ? 0 : SKGC_E_HIP;
}

int skgc_layer_norm(const float* x, const float* gamma, const float* beta, const float* res, float* out, long long rows, int N, void* stream) {
    if (!x || !gamma || !beta || !out || rows <= 0 || N <= 0 || N > 1024) return SKGC_E_ARG;
    hipLaunchKernelGGL(layer_norm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), x, gamma, beta, res, out, rows, N);
    return hipGetLastError() == hipSuccess ? 0 : SKGC_E_HIP;
}

int skgc_segment_sum(const float* e, const int* offsets, float* out, float* acc, int n_nodes, int N, void* stream) {
    if (!e || !offsets || !out || n_nodes <= 0 || N <= 0 || (N & 3)) return SKGC_E_ARG;
    hipLaunchKernelGGL(segment_sum_kernel, dim3((unsigned)((n_nodes + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), e, offsets, out, acc, n_nodes, N);
    return hipGetLastError() == hipSuccess ? 0 : SKGC_E_HIP;
}

int skgc_prepare_weight_perm8(const float* src, int N, int K, void* dst, long long plane, int ldw, void* stream) {
    if (!src || !dst || N <= 0 || K <= 0 || (N & 31) || ldw < K || (ldw & 7) || plane < (long long)N * ldw) return SKGC_E_ARG;
    const hipError_t e = prep_weight<f16, 2>(src, static_cast<f16*>(dst), plane, N, K, ldw, K, 1, 0, 1, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? 0 : SKGC_E_HIP;
}

int skgc_linear_layer_norm(const float* a, long long lda, int K, const void* w, long long w_plane, int ldw, const float* bias, const float* gamma,
                           const float* beta, const float* res, float* out, long long rows, void* stream) {
    if (!a || !w || !gamma || !beta || !out || rows <= 0 || rows > 0x7fffffff || K <= 0 || lda < K || (ldw & 7) || ldw < K) return SKGC_E_ARG;
    if (ln_tile_rows() == 64) return linear_layer_norm_t<TLN64>(a, lda, K, w, w_plane, ldw, bias, gamma, beta, res, out, rows, stream);
    return linear_layer_norm_t<TLN128>(a, lda, K, w, w_plane, ldw, bias, gamma, beta, res, out, rows, stream);
}

int skgc_sum_linear_layer_norm(const skgc_sum_desc* d, void* stream) {
    if (!d || !d->w || !d->gamma || !d->beta || !d->out || d->rows <= 0 || d->rows > 0x7fffffff || d->K <= 0 || (d->K & 7) || d->n_src < 1 || d->n_src > 3 ||
        (d->ldw & 7) || d->ldw < d->K || (d->act != 0 && d->act != 2))
        return SKGC_E_ARG;
    ALSumGather al;
    for (int s = 0; s < 3; ++s) {
        al.src[s] = nullptr; al.idx[s] = nullptr; al.ld[s] = 0;
        if (s < d->n_src) {
            if (!d->src[s] || d->ld[s] < d->K || (d->ld[s] & 3) || (reinterpret_cast<size_t>(d->src[s]) & 15)) return SKGC_E_ARG;
            al.src[s] = d->src[s]; al.idx[s] = d->idx[s]; al.ld[s] = d->ld[s];
        }
    }
    al.n_src = d->n_src; al.M = (int)d->rows; al.K = d->K; al.act = d->act;
    if (d->group == 3) {            // every source indexed (the index arrays define the virtual row order), no residual
        for (int s = 0; s < d->n_src; ++s)
            if (!d->idx[s]) return SKGC_E_ARG;
        if (d->res) return SKGC_E_ARG;
        return sum3_linear_layer_norm(d, al, stream);
    }
    if (d->group != 0 && d->group != 1) return SKGC_E_ARG;
    if (ln_tile_rows() == 64) return sum_linear_layer_norm_t<TLN64>(d, al, stream);
    return sum_linear_layer_norm_t<TLN128>(d, al, stream);
}

}  // extern "C"
