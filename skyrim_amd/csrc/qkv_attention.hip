// QKV linear + earth-specific window attention in ONE kernel: Q, K, V never exist in HBM.
//
// The two-kernel path writes Q/K/V (6.4 GB/step) through the CUs' 8 B/clk store path and reads them back; here a
// workgroup is (window, group of 6 heads): its 6 waves share the window's 144 input rows in LDS (hi plane of the residual
// stream, gathered through the window table = pad + roll + partition), each wave computes ITS head's Q, K and V with
// the weight fragments streamed from L2 straight into registers, and keeps them as fp16 MFMA fragments for the
// attention core of attention.hip.  Modes with a single-plane A operand for the QKV linear (f16x3q, f16x3qh, f16).
//
// The accumulator layouts of the three products ARE the fragment layouts the attention MFMAs want, given the row orders
// (no data movement, no LDS round trip for Q/K/V):
//   Q, K:  D^T = W X^T (W fragment as the A operand): lane (token l&15, rows 4g+r of the perm8-ordered W fragment pair)
//          = head-dim 8g..8g+7 of one token = B operand (Q) / A operand (K) of S^T = K Q^T.  K's X rows are read in
//          attention.hip's key order attn_key(), Q's in natural order.
//   V:     D = X W^T (X fragment as the A operand, rows in key order): lane (prepared W row l&15 -> head-dim
//          8(l15>>2) + 4b + (l15&3), tokens 4g+r of fragments 2kb, 2kb+1 = keys 32kb + 8g + [0..7]) = A operand V^T of
//          O^T = V^T P^T, with the same head-dim row order as attention.hip.
#include <type_traits>
#include "common.h"
#include "gemm_dma.h"
#include "launchers.h"

namespace skp {

__device__ __forceinline__ int attn_key_f(int f, int row) {      // attention.hip: key held by row `row` of K fragment f
    return f < 8 ? 32 * (f >> 1) + 8 * (row >> 2) + 4 * (f & 1) + (row & 3) : 128 + row;
}

struct QkvAttnArgs {
    const f16* xs;          // residual stream, hi plane, blocked layout [tokens][C]
    const int* widx;        // window gather table (n_win * 144 entries, -1 = padding)
    const f16* W;           // prepared QKV weight [3C][C], blocked, perm8 row order; lo plane at + w_plane
    long long w_plane;
    const float* bias;      // [3C]
    const f16* bias_exp;    // expanded earth-specific bias (+ mask), attention.hip layout
    const f16* zrow;
    void* out;              // attention output planes, blocked layout [n_win*144][C]
    long long out_plane;
    int n_win, nW, heads, groups;
    float scale;
};

constexpr int kFusedWaves = 6;
constexpr int kXChunk = WIN_TOKENS * 64;     // LDS bytes of one 32-column chunk of the window's rows

template <class TO, int NPL_O, int C, int NW>
__global__ void __launch_bounds__(64 * kFusedWaves) fused_qkv_attention_kernel(const QkvAttnArgs g) {
    constexpr int NK = C / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, gq = lane >> 4;
    const int win = blockIdx.x / g.groups, hg = blockIdx.x - win * g.groups;
    const int head = hg * kFusedWaves + wave;
    const int type = win / g.nW;

    // ---- the window's rows (hi plane) -> LDS, [chunk][144 rows][64 B], slots swizzled as in gemm_dma.h ---- //
    {
        const unsigned lds_base = (unsigned)(size_t)smem;
        const int lr = lane >> 2, lp = lane & 3;
        constexpr int PIECES = NK * 9;
#pragma unroll 1
        for (int q = wave; q < PIECES; q += kFusedWaves) {
            const int kc = q / 9, p = q - kc * 9;
            const int row = p * 16 + lr;
            const int tok = g.widx[win * WIN_TOKENS + row];
            const int chunk = lp ^ ((row >> 1) & 3);
            const f16* src = tok >= 0 ? g.xs + blk_off(tok, kc * 32 + chunk * 8, C) : g.zrow;
            glds16(src, lds_base + (unsigned)(kc * kXChunk + p * 1024));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- one head's Q / K / V: acc[a][b] over K = C, W fragments double-buffered from L2 ---- //
    f32x4 acc[9][2];
    auto qkv_pass = [&](auto perm_c, auto w_is_a_c, int n0) {
        constexpr bool PERM = decltype(perm_c)::value, W_IS_A = decltype(w_is_a_c)::value;
#pragma unroll
        for (int a = 0; a < 9; ++a) { acc[a][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[a][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        const f16* wp = g.W + (long long)((n0 >> 4) * NK) * 512 + l15 * 32 + gq * 8;      // fragment b at + b * NK * 512, chunk kc at + kc * 512
        uint4 wn[2][NW];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int p = 0; p < NW; ++p) wn[b][p] = *reinterpret_cast<const uint4*>(wp + (long long)b * NK * 512 + p * g.w_plane);
#pragma unroll 1
        for (int kc = 0; kc < NK; ++kc) {
            uint4 wc[2][NW];
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int p = 0; p < NW; ++p) wc[b][p] = wn[b][p];
            if (kc + 1 < NK) {
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int p = 0; p < NW; ++p)
                        wn[b][p] = *reinterpret_cast<const uint4*>(wp + ((long long)b * NK + kc + 1) * 512 + p * g.w_plane);
            }
            const char* xc = smem + kc * kXChunk;
#pragma unroll
            for (int a = 0; a < 9; ++a) {
                const int row = PERM ? attn_key_f(a, l15) : a * 16 + l15;
                const uint4 xa = *reinterpret_cast<const uint4*>(xc + lds_off<32>(row, gq));
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    if constexpr (W_IS_A) {
                        if constexpr (NW == 2) acc[a][b] = OpT<f16>::mfma(as_v8<f16>(wc[b][1]), as_v8<f16>(xa), acc[a][b]);
                        acc[a][b] = OpT<f16>::mfma(as_v8<f16>(wc[b][0]), as_v8<f16>(xa), acc[a][b]);
                    } else {
                        if constexpr (NW == 2) acc[a][b] = OpT<f16>::mfma(as_v8<f16>(xa), as_v8<f16>(wc[b][1]), acc[a][b]);
                        acc[a][b] = OpT<f16>::mfma(as_v8<f16>(xa), as_v8<f16>(wc[b][0]), acc[a][b]);
                    }
                }
            }
        }
    };
    auto pack_rows = [&](const float4& b0, const float4& b1, float s, uint4 (&dst)[9]) {      // Q / K: 8 head-dims of one token
#pragma unroll
        for (int a = 0; a < 9; ++a) {
            const float v[8] = {(acc[a][0][0] + b0.x) * s, (acc[a][0][1] + b0.y) * s, (acc[a][0][2] + b0.z) * s, (acc[a][0][3] + b0.w) * s,
                                (acc[a][1][0] + b1.x) * s, (acc[a][1][1] + b1.y) * s, (acc[a][1][2] + b1.z) * s, (acc[a][1][3] + b1.w) * s};
            uint4 o[1];
            split8<f16, 1>(v, o);
            dst[a] = o[0];
        }
    };

    uint4 qfr[9], kf[9], vf[2][5];
    {
        const float* bq = g.bias + head * HEAD_DIM + gq * 8;
        qkv_pass(std::false_type{}, std::true_type{}, head * HEAD_DIM);
        pack_rows(*reinterpret_cast<const float4*>(bq), *reinterpret_cast<const float4*>(bq + 4), g.scale, qfr);
        const float* bk = g.bias + C + head * HEAD_DIM + gq * 8;
        qkv_pass(std::true_type{}, std::true_type{}, C + head * HEAD_DIM);
        pack_rows(*reinterpret_cast<const float4*>(bk), *reinterpret_cast<const float4*>(bk + 4), 1.0f, kf);
        qkv_pass(std::true_type{}, std::false_type{}, 2 * C + head * HEAD_DIM);
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const float bv = g.bias[2 * C + head * HEAD_DIM + perm8_col(16 * b + l15)];
#pragma unroll
            for (int kb = 0; kb < 5; ++kb) {
                float v[8];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = acc[2 * kb][b][r] + bv;
                    v[4 + r] = kb < 4 ? acc[(2 * kb + 1) % 9][b][r] + bv : 0.f;
                }
                uint4 o[1];
                split8<f16, 1>(v, o);
                vf[b][kb] = o[0];
            }
        }
    }

    // ---- attention core (attention.hip), Q/K/V from registers ---- //
    const f16* bp = g.bias_exp + ((long long)type * g.heads + head) * (81 * 256);
    TO* out = reinterpret_cast<TO*>(g.out);
    const float LOG2E = 1.4426950408889634f;
    uint4 bn[4];
    uint2 bn8;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) bn[kb] = *reinterpret_cast<const uint4*>(bp + kb * 512 + lane * 8);
    bn8 = *reinterpret_cast<const uint2*>(bp + 2048 + lane * 4);
#pragma unroll
    for (int qf = 0; qf < 9; ++qf) {
        uint4 bcur[4];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) bcur[kb] = bn[kb];
        const uint2 bcur8 = bn8;
        if (qf < 8) {
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) bn[kb] = *reinterpret_cast<const uint4*>(bp + (qf + 1) * 2304 + kb * 512 + lane * 8);
            bn8 = *reinterpret_cast<const uint2*>(bp + (qf + 1) * 2304 + 2048 + lane * 4);
        }
        f32x4 s[9];
#pragma unroll
        for (int f = 0; f < 9; ++f) s[f] = OpT<f16>::mfma(as_v8<f16>(kf[f]), as_v8<f16>(qfr[qf]), f32x4{0.f, 0.f, 0.f, 0.f});
        float mx = -3.0e38f;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const typename OpT<f16>::v8 b = as_v8<f16>(bcur[kb]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s[2 * kb][r] += (float)b[r];
                s[2 * kb + 1][r] += (float)b[4 + r];
                mx = fmaxf(mx, fmaxf(s[2 * kb][r], s[2 * kb + 1][r]));
            }
        }
        {
            typedef f16 h4 __attribute__((ext_vector_type(4)));
            const h4 b = __builtin_bit_cast(h4, bcur8);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s[8][r] += (float)b[r];
                mx = fmaxf(mx, s[8][r]);
            }
        }
        mx = fmaxf(mx, __shfl_xor(mx, 16));
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
        const float mxl = mx * LOG2E;
#pragma unroll
        for (int f = 0; f < 9; ++f)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = exp2f(s[f][r] * LOG2E - mxl);
                s[f][r] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        const float inv = 1.0f / sum;
        f32x4 o[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int kb = 0; kb < 5; ++kb) {
            float pv[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pv[r] = s[2 * kb][r];
                pv[4 + r] = (kb < 4) ? s[(2 * kb + 1) % 9][r] : 0.f;
            }
            uint4 pf[1];
            split8<f16, 1>(pv, pf);
#pragma unroll
            for (int df = 0; df < 2; ++df) o[df] = OpT<f16>::mfma(as_v8<f16>(vf[df][kb]), as_v8<f16>(pf[0]), o[df]);
        }
        TO* orow = out + blk_off((long long)win * WIN_TOKENS + qf * 16 + l15, head * HEAD_DIM, C) + gq * 8;
        const float y[8] = {o[0][0] * inv, o[0][1] * inv, o[0][2] * inv, o[0][3] * inv, o[1][0] * inv, o[1][1] * inv, o[1][2] * inv, o[1][3] * inv};
        store8_planes<TO, NPL_O>(orow, g.out_plane, y);
    }
}

template <class TO, int NPL_O, int NW>
hipError_t launch_fused_qkv_attention(const QkvAttnArgs& a, int C, hipStream_t stream) {
    if (a.heads % kFusedWaves != 0) return hipErrorInvalidValue;
    const unsigned grid = (unsigned)(a.n_win * a.groups);
    if (C == 192) {
        auto kern = fused_qkv_attention_kernel<TO, NPL_O, 192, NW>;
        constexpr int smem = 6 * kXChunk;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * kFusedWaves), smem, stream, a);
    } else if (C == 384) {
        auto kern = fused_qkv_attention_kernel<TO, NPL_O, 384, NW>;
        constexpr int smem = 12 * kXChunk;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * kFusedWaves), smem, stream, a);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// op_qkv + launch_attention in one launch (modes whose QKV linear reads one fp16 plane of the stream)
template <class P>
hipError_t op_qkv_attention(const Geom& g, const BlockW<typename P::T>& b, const int* widx, int res, const typename P::T* Xs, const Work<P>& wk, hipStream_t s) {
    if constexpr (!std::is_same<typename P::T, f16>::value) {
        return hipErrorInvalidValue;
    } else {
        const int C = res == 0 ? 192 : 384, heads = C / HEAD_DIM;
        QkvAttnArgs a{Xs, widx, b.qkv.w, b.qkv.plane, b.qkv_b, b.bias_exp, wk.zrow, wk.ao, wk.ao_plane,
                      g.nwin[res], g.nW[res], heads, heads / kFusedWaves, 0.17677669529663687f};
        constexpr int NPL_O = (P::NA > P::NW ? P::NA : P::NW);
        return launch_fused_qkv_attention<f16, NPL_O, P::NW>(a, C, s);
    }
}

template hipError_t op_qkv_attention<PrecBF16x3>(const Geom&, const BlockW<bf16>&, const int*, int, const bf16*, const Work<PrecBF16x3>&, hipStream_t);
template hipError_t op_qkv_attention<PrecF16>(const Geom&, const BlockW<f16>&, const int*, int, const f16*, const Work<PrecF16>&, hipStream_t);
template hipError_t op_qkv_attention<PrecF16x3>(const Geom&, const BlockW<f16>&, const int*, int, const f16*, const Work<PrecF16x3>&, hipStream_t);

}  // namespace skp
