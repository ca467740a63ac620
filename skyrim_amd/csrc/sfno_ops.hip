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

#include <cstdlib>

#include "../../include/skyrim_sfno.h"
#include "strided_gemm.h"

namespace skp {



// Two tiles.  TG (128 x 256, strided_gemm.h): 64 x 64 wave tiles, 162-177 VGPRs, ONE 8-wave workgroup per CU.  TS (128 x 128): 64 x 32 wave
// tiles fit 128 VGPRs, so TWO 8-wave workgroups (16 waves) share a CU and one's loads / stores run under the other's MFMAs -- measured
// a little faster for the DFTs (-9 %), the synthesis (-9 %) and dhconv (-5 %), slower for the Legendre analysis (N = 240 fits one 256-wide
// tile: +15 %), which therefore keeps TG; 17.8 -> 17.6 ms/step in all (SKSFNO_TILE=256 / 128 force one tile).
typedef TileCfg<128, 128, 32, 2, 4> TS;

template <class PX, class AL, bool SWAP, class TC>
__device__ __forceinline__ void gemm_strided_body(GemmArgs<PX, AL, EpStrided>& g, const BatchStrides& bs) {
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
    // Which tile this workgroup computes.  Workgroups go to the 8 XCDs round-robin in launch order (x fastest), so with the plain mapping the
    // column tiles of one row tile -- which all read the same 128 rows of A -- land on 8 different L2s and every XCD streams the whole of A
    // (GraphCast's node-term GEMM, N = 1024 = 8 column tiles: 1.44 GB fetched per launch for an 84 MB operand).  Re-mapped, the workgroups
    // of one XCD (same launch index mod 8) walk a contiguous range of tiles, column tile fastest: a row tile's columns run side by side on one
    // L2.  The map is a bijection for any tile count (the first T mod 8 XCDs take one tile more).
    int tx = blockIdx.x, ty = blockIdx.y;
    if (bs.xcd_remap) {
        const int gx = gridDim.x, T = gx * (int)gridDim.y, l = tx + gx * ty;
        const int q = T >> 3, r = T & 7, x = l & 7, slot = l >> 3;
        const int v = x < r ? x * (q + 1) + slot : r * (q + 1) + (x - r) * q + slot;
        ty = v / gx;
        tx = v - ty * gx;
    }
    if (bs.m_cap_step > 0) {
        const int cap = bs.m_cap0 + (int)z * bs.m_cap_step;
        if (cap < g.M) { g.M = cap; g.al.M = cap; }
        if (ty * TC::BM >= g.M) return;
    }
    gemm_body<PX, TC, AL, EpStrided, SWAP>(g, smem, tx, ty);
}

template <class PX, class AL, bool SWAP>
__global__ void __launch_bounds__(TG::THREADS) gemm_strided_kernel(GemmArgs<PX, AL, EpStrided> g, BatchStrides bs) {
    gemm_strided_body<PX, AL, SWAP, TG>(g, bs);
}

template <class PX, class AL, bool SWAP>
__global__ void __launch_bounds__(TS::THREADS) __attribute__((amdgpu_waves_per_eu(4, 4))) gemm_strided_kernel_s(GemmArgs<PX, AL, EpStrided> g, BatchStrides bs) {
    gemm_strided_body<PX, AL, SWAP, TS>(g, bs);
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
        (d->ldw & 7) || d->ldw < d->K || (d->act < 0 || d->act > 2) || d->k_lo_step < 0 || d->m_cap_step < 0 || (d->terms != 0 && d->terms != 2 && d->terms != 3) ||
        (d->a_kscale == nullptr) != (d->a_kshift == nullptr) || (d->a_kscale != nullptr && d->k_lo_step > 0) ||
        (d->a2 != nullptr && (d->a2_k_split <= 0 || (d->a2_k_split & 7) || d->a2_k_split >= d->K || d->k_lo_step > 0 || d->batch != 1)))
        return SKSFNO_E_ARG;
    const ALStrided al{d->a, d->M, d->K, d->a_m1, d->a_sm, d->a_sm2, d->a_sk, d->a_kscale, d->a_kshift, d->a2, d->a2_sk, d->a2_k_split};
    const EpStrided ep{d->out, d->bias, d->res_pre, d->res_post, d->o_m1, d->act, d->o_sm, d->o_sm2, d->o_sn};
    static const bool no_remap = getenv("SKSFNO_NO_XCD_REMAP") != nullptr;                                     // A/B switch (tools/r4_remap.sh)
    const BatchStrides bs{d->a_sb, d->w_sb, d->o_sb, d->k_lo_step, d->m_cap0, d->m_cap_step, no_remap ? 0 : 1};
    // rows contiguous in the output (NCHW activations): un-swapped order gives 4 consecutive rows per lane
    const bool swap = !(d->o_sm == 1 && d->o_sn != 1);
    static const int tile_env = [] { const char* v = getenv("SKSFNO_TILE"); return v ? atoi(v) : 0; }();      // 128 / 256: force one tile
    const bool wide = tile_env == 256 || (tile_env != 128 && !swap && d->N > 128 && d->N <= 256);            // one 256-wide tile covers N
    const int bn = wide ? TG::BN : TS::BN;
    const dim3 grid((d->N + bn - 1) / bn, (d->M + TG::BM - 1) / TG::BM, d->batch);
    static_assert(TG::BM == TS::BM && TG::THREADS == TS::THREADS, "the two tiles share the row split and the block size");
    if (grid.y > 65535 || grid.z > 65535) return SKSFNO_E_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    auto launch = [&](auto px, auto loader) {
        typedef decltype(px) PX;
        typedef decltype(loader) AL;
        GemmArgs<PX, AL, EpStrided> g;
        g.al = loader; g.ep = ep;
        g.W = static_cast<const f16*>(d->w);
        g.w_plane = d->w_plane;
        g.ldw = d->ldw;
        g.M = d->M; g.N = d->N; g.K = d->K;
        constexpr int smem = gemm_smem_bytes<PX, TG>() + kEpiScratch;
        if constexpr (smem > 64 * 1024) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_strided_kernel<PX, AL, true>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_strided_kernel<PX, AL, false>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        }
        constexpr int smem_s = gemm_smem_bytes<PX, TS>() + kEpiScratch;
        if (wide) {
            if (swap) hipLaunchKernelGGL((gemm_strided_kernel<PX, AL, true>), grid, dim3(TG::THREADS), smem, st, g, bs);
            else      hipLaunchKernelGGL((gemm_strided_kernel<PX, AL, false>), grid, dim3(TG::THREADS), smem, st, g, bs);
        } else {
            if (swap) hipLaunchKernelGGL((gemm_strided_kernel_s<PX, AL, true>), grid, dim3(TS::THREADS), smem_s, st, g, bs);
            else      hipLaunchKernelGGL((gemm_strided_kernel_s<PX, AL, false>), grid, dim3(TS::THREADS), smem_s, st, g, bs);
        }
    };
    // the loader without per-element predicates wherever the operand allows it (strided_gemm.h: ALFast)
    const long long a_extent = (long long)((d->M - 1) / d->a_m1) * d->a_sm2 + (long long)(d->a_m1 < d->M ? d->a_m1 - 1 : d->M - 1) * d->a_sm + (long long)(d->K - 1) * d->a_sk;
    static const bool no_fast = getenv("SKSFNO_NO_FAST_LOADER") != nullptr;
    const bool fast = !no_fast && d->terms != 2 && !d->a_kscale && !d->a2 && (d->K & 7) == 0 && d->a_sm >= 0 && d->a_sm2 >= 0 && d->a_sk > 0 && a_extent < (1ll << 30);
    const bool vec = fast && d->a_sk == 1 && (reinterpret_cast<size_t>(d->a) & 15) == 0 && !(d->a_sm & 3) && !(d->a_sm2 & 3) && !(d->a_sb & 3);
    if (vec)       launch(PrecF16x3{}, ALFast<true>{d->a, d->M, d->K, d->a_m1, d->a_sm, d->a_sm2, d->a_sk});
    else if (fast) launch(PrecF16x3{}, ALFast<false>{d->a, d->M, d->K, d->a_m1, d->a_sm, d->a_sm2, d->a_sk});
    else if (d->terms == 2) launch(PrecF16x2W{}, al);      // A as one fp16 plane, W hi/lo
    else                    launch(PrecF16x3{}, al);
    return hipGetLastError() == hipSuccess ? 0 : SKSFNO_E_HIP;
}

int sksfno_instance_norm(const float* x, const float* gamma, const float* beta, float* out, int C, long long HW, float eps, void* stream) {
    if (!x || !gamma || !beta || !out || C <= 0 || HW <= 0) return SKSFNO_E_ARG;
    hipLaunchKernelGGL(instance_norm_kernel, dim3(C), dim3(1024), 0, static_cast<hipStream_t>(stream), x, gamma, beta, out, HW, eps);
    return hipGetLastError() == hipSuccess ? 0 : SKSFNO_E_HIP;
}

}  // extern "C"
