// SFNO (FourCastNet v2) building blocks behind include/skyrim_sfno.h: every linear map of the network -- 1x1
// convolutions, the truncated real DFT along longitude, the Legendre analysis / synthesis per order m, the per-degree
// complex channel mixing (dhconv) and their inverses -- is ONE batched GEMM against a constant matrix:
//
//     out[b](m, n) = post( act( sum_k A[b](m, k) * W[b][n][k] + bias[n] + res_pre[b](m, n) ) ) + res_post[b](m, n)
//
// A and out are fp32 arrays addressed through strides (two-level for the row index, so that (channel, re/im) pairs and
// NCHW / channel-last layouts need no transpose pass); W is prepared once as fp16 hi/lo planes and the product runs as
// three fp16 MFMA terms with fp32 accumulation (fp32-class results at 1/3 of the MFMA rate).  The main
// loop is gemm.h's register-staged pipeline (the A operand is converted on the fly).  First correct path: the strided
// loader issues scalar loads when k is not the contiguous index; layout-specialised loaders are the next step.
#include <hip/hip_runtime.h>

#include "../../include/skyrim_sfno.h"
#include "common.h"
#include "epilogues.h"
#include "gemm.h"
#include "launchers.h"

namespace skp {

// ---- A operand: fp32, strided ------------------------------------------------------------------ //
struct ALStrided {
    static constexpr bool kDirect = false;
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

typedef PrecF16x3 PG;
typedef TileCfg<128, 256, 32, 2, 4> TG;      // wide N: the fp32 A operand is fetched and split once per 256 output columns

struct BatchStrides { long long a, w, o; int k_lo_step, m_cap0, m_cap_step; };

template <class PX, bool SWAP>
__global__ void __launch_bounds__(TG::THREADS) gemm_strided_kernel(GemmArgs<PX, ALStrided, EpStrided> g, BatchStrides bs) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const long long z = blockIdx.z;
    g.al.a += z * bs.a;
    g.W += z * bs.w;
    g.ep.out += z * bs.o;
    if (g.ep.res_pre) g.ep.res_pre += z * bs.o;
    if (g.ep.res_post) g.ep.res_post += z * bs.o;
    if (bs.k_lo_step > 0) {                       // contraction starts at the first 32-aligned k that can be non-zero
        const int kb = ((int)z * bs.k_lo_step) & ~31;
        if (kb >= g.K) return;
        g.al.a += (long long)kb * g.al.sk;
        g.W += kb;
        g.K -= kb;
        g.al.K -= kb;
    }
    if (bs.m_cap_step > 0) {
        const int cap = bs.m_cap0 + (int)z * bs.m_cap_step;
        if (cap < g.M) { g.M = cap; g.al.M = cap; }
        if ((int)blockIdx.y * TG::BM >= g.M) return;
    }
    gemm_body<PX, TG, ALStrided, EpStrided, SWAP>(g, smem);
}

// ---- instance norm over (H, W) per channel: two passes for the statistics, one to apply ---- //
__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    float s = 0.f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
    return s;
}

__global__ void __launch_bounds__(1024) instance_norm_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float* __restrict__ out, long long HW, float eps) {
    __shared__ float red[16];
    const float* xc = x + (long long)blockIdx.x * HW;
    float* oc = out + (long long)blockIdx.x * HW;
    const bool v4 = (HW & 3) == 0;
    // one pass for both moments, shifted by the channel's first value (var = E[(x-p)^2] - E[x-p]^2 keeps its digits as long as
    // |mean - p| is a few standard deviations): 2 reads + 1 write of the tensor instead of 3 + 1
    const float pv = xc[0];
    float s = 0.f, q = 0.f;
    if (v4) {
        for (long long i = threadIdx.x; i < HW / 4; i += blockDim.x) {
            const float4 v = reinterpret_cast<const float4*>(xc)[i];
            const float a = v.x - pv, b = v.y - pv, c = v.z - pv, d = v.w - pv;
            s += (a + b) + (c + d);
            q += (a * a + b * b) + (c * c + d * d);
        }
    } else {
        for (long long i = threadIdx.x; i < HW; i += blockDim.x) { const float d = xc[i] - pv; s += d; q += d * d; }
    }
    const float m1 = block_sum(s, red) / (float)HW;
    const float m2 = block_sum(q, red) / (float)HW;
    const float mean = pv + m1;
    const float rstd = rsqrtf(fmaxf(m2 - m1 * m1, 0.f) + eps);
    const float g = gamma[blockIdx.x] * rstd, b = beta[blockIdx.x] - mean * g;
    if (v4) {
        for (long long i = threadIdx.x; i < HW / 4; i += blockDim.x) {
            const float4 v = reinterpret_cast<const float4*>(xc)[i];
            reinterpret_cast<float4*>(oc)[i] = make_float4(v.x * g + b, v.y * g + b, v.z * g + b, v.w * g + b);
        }
    } else {
        for (long long i = threadIdx.x; i < HW; i += blockDim.x) oc[i] = xc[i] * g + b;
    }
}

}  // namespace skp

using namespace skp;

extern "C" {

int sksfno_abi_version(void) { return SKSFNO_ABI_VERSION; }

int sksfno_prepare_weight(const float* src, long long sn, long long sk, int N, int K, void* dst, long long plane, int ldw, void* stream) {
    if (!src || !dst || N <= 0 || K <= 0 || ldw < K || (ldw & 7) || plane < (long long)N * ldw) return SKSFNO_E_ARG;
    const hipError_t e = prep_weight<f16, 2>(src, static_cast<f16*>(dst), plane, N, K, ldw, sn, sk, 0, 0, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? 0 : SKSFNO_E_HIP;
}

int sksfno_gemm_run(const sksfno_gemm* d, void* stream) {
    if (!d || !d->a || !d->w || !d->out || d->M <= 0 || d->N <= 0 || d->K <= 0 || d->batch <= 0 || d->a_m1 <= 0 || d->o_m1 <= 0 ||
        (d->ldw & 7) || d->ldw < d->K || (d->act != 0 && d->act != 1) || d->k_lo_step < 0 || d->m_cap_step < 0 || (d->terms != 0 && d->terms != 2 && d->terms != 3) ||
        (d->a_kscale == nullptr) != (d->a_kshift == nullptr) || (d->a_kscale != nullptr && d->k_lo_step > 0) ||
        (d->a2 != nullptr && (d->a2_k_split <= 0 || (d->a2_k_split & 7) || d->a2_k_split >= d->K || d->k_lo_step > 0 || d->batch != 1)))
        return SKSFNO_E_ARG;
    const ALStrided al{d->a, d->M, d->K, d->a_m1, d->a_sm, d->a_sm2, d->a_sk, d->a_kscale, d->a_kshift, d->a2, d->a2_sk, d->a2_k_split};
    const EpStrided ep{d->out, d->bias, d->res_pre, d->res_post, d->o_m1, d->act, d->o_sm, d->o_sm2, d->o_sn};
    const BatchStrides bs{d->a_sb, d->w_sb, d->o_sb, d->k_lo_step, d->m_cap0, d->m_cap_step};
    const dim3 grid((d->N + TG::BN - 1) / TG::BN, (d->M + TG::BM - 1) / TG::BM, d->batch);
    if (grid.y > 65535 || grid.z > 65535) return SKSFNO_E_ARG;
    // rows contiguous in the output (NCHW activations): un-swapped order gives 4 consecutive rows per lane
    const bool swap = !(d->o_sm == 1 && d->o_sn != 1);
    hipStream_t st = static_cast<hipStream_t>(stream);
    auto launch = [&](auto px) {
        typedef decltype(px) PX;
        GemmArgs<PX, ALStrided, EpStrided> g;
        g.al = al; g.ep = ep;
        g.W = static_cast<const f16*>(d->w);
        g.w_plane = d->w_plane;
        g.ldw = d->ldw;
        g.M = d->M; g.N = d->N; g.K = d->K;
        constexpr int smem = gemm_smem_bytes<PX, TG>() + kEpiScratch;
        if (swap) hipLaunchKernelGGL((gemm_strided_kernel<PX, true>), grid, dim3(TG::THREADS), smem, st, g, bs);
        else      hipLaunchKernelGGL((gemm_strided_kernel<PX, false>), grid, dim3(TG::THREADS), smem, st, g, bs);
    };
    if (d->terms == 2) launch(PrecF16x2W{});      // A as one fp16 plane, W hi/lo
    else               launch(PrecF16x3{});
    return hipGetLastError() == hipSuccess ? 0 : SKSFNO_E_HIP;
}

int sksfno_instance_norm(const float* x, const float* gamma, const float* beta, float* out, int C, long long HW, float eps, void* stream) {
    if (!x || !gamma || !beta || !out || C <= 0 || HW <= 0) return SKSFNO_E_ARG;
    hipLaunchKernelGGL(instance_norm_kernel, dim3(C), dim3(1024), 0, static_cast<hipStream_t>(stream), x, gamma, beta, out, HW, eps);
    return hipGetLastError() == hipSuccess ? 0 : SKSFNO_E_HIP;
}

}  // extern "C"
