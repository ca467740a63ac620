// Earth-specific 3D window attention (EarthAttention3D core) for gfx950.
//
// One wavefront = one (window, head): 144 tokens x head_dim 32.  Everything a wave needs is read
// straight into MFMA fragment registers -- Q/K rows are 64-byte contiguous, V arrives transposed
// ([d][token], written that way by the QKV epilogue) so each lane's 4 keys are 8 contiguous bytes,
// and the earth-specific bias (+ shifted-window mask) is pre-expanded per (window type, head) in
// exactly the accumulator layout.  No LDS, no barriers; the only cross-lane traffic is the 2-step
// butterfly of the softmax row reduction.
//
//   S^T[key][q] = K Q^T   (A = K frag, B = Q frag)   -> lane (q = l&15) holds 4 keys per fragment
//   O^T[d][q]   = V^T P^T (A = V^T frag, B = P frag) -> lane (q = l&15) holds 4 head-dim rows per fragment
//
// Which K row / V^T row sits in which fragment row is free (it is only an address), and is chosen so that every
// HBM/L2 access of the kernel is 16 bytes per lane (8-byte accesses cost the same per wave instruction and therefore
// twice per byte; the CU's memory pipeline is what bounds this kernel -- DESIGN.md 5.2):
//   * key blocks kb = 0..3: row 4g+i of K fragment (2kb + jj) is key 32kb + 8g + 4jj + i, so a lane's accumulators of the
//     fragment pair are the 8 CONSECUTIVE keys 32kb + 8g + [0..7] = exactly MFMA k-slots 8g..8g+7 of the PV product:
//     the softmax output feeds the second MFMA without data movement, the V^T fragment is one 16-byte load of 8
//     consecutive tokens, and the bias fragment pair is one 16-byte load; fragment 8 (keys 128..143) keeps the plain
//     order 128 + 4g + i with 8-byte accesses;
//   * row 4g+i of V^T fragment df is head-dim row 8g + 4df + i, so a lane's two output accumulators are 8 consecutive
//     columns of the attention output: one 16-byte store per plane.
#include "common.h"
#include "launchers.h"

namespace skp {

// key held by fragment row `row` (= 4g + i) of K fragment f (see the header comment)
__device__ __forceinline__ int attn_key(int f, int row) {
    return f < 8 ? 32 * (f >> 1) + 8 * (row >> 2) + 4 * (f & 1) + (row & 3) : 128 + row;
}

// Q, K, V and the softmax output P are single fp16 planes in EVERY precision mode: attention is the least
// rounding-sensitive part of the network (measured on the oracle: fp16 q/k/v -> 7e-5, fp16 P -> 4e-5 per-channel
// error, vs 3.3e-4 / 3.6e-4 for the attention output / MLP hidden, which therefore stay hi/lo split), so
// QK^T and PV are one MFMA term each and the Q/K/V round trip through HBM is 2 bytes per element.
// 165 VGPRs -> three waves per SIMD (the default schedule takes 189 = two; four would spill ~40 registers)
template <class TO, int NPL_O>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(3, 3))) earth_attention_kernel(const f16* __restrict__ q, const f16* __restrict__ k,
                                                              const f16* __restrict__ vt, long long plane,
                                                              const f16* __restrict__ bias_exp, TO* __restrict__ out, long long out_plane,
                                                              int ld_out, int n_win, int nW, int heads) {
    typedef f16 T;
    constexpr int NPL = 1;
    const int lane = threadIdx.x & 63;
    const long long wg = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wg >= (long long)n_win * heads) return;
    // consecutive waves share (type, head) -> the expanded bias tile stays hot in L2
    const int wi = (int)(wg % nW);
    const int th = (int)(wg / nW);
    const int head = th % heads, type = th / heads;
    const int win = type * nW + wi;
    const long long base = (long long)win * heads + head;
    const int l15 = lane & 15, g = lane >> 4;

    const T* qp = q + base * (WIN_TOKENS * HEAD_DIM) + l15 * HEAD_DIM + g * 8;
    const T* kp = k + base * (WIN_TOKENS * HEAD_DIM) + g * 8;
    const T* vp = vt + base * (WIN_TOKENS * HEAD_DIM) + (8 * (l15 >> 2) + (l15 & 3)) * WIN_TOKENS;      // + 4 df rows
    const f16* bp = bias_exp + ((long long)type * heads + head) * (81 * 256);

    uint4 kf[NPL][9];
#pragma unroll
    for (int p = 0; p < NPL; ++p)
#pragma unroll
        for (int f = 0; f < 9; ++f) kf[p][f] = *reinterpret_cast<const uint4*>(kp + p * plane + attn_key(f, l15) * HEAD_DIM);

    uint4 vf[NPL][2][5];
#pragma unroll
    for (int p = 0; p < NPL; ++p)
#pragma unroll
        for (int df = 0; df < 2; ++df)
#pragma unroll
            for (int kb = 0; kb < 5; ++kb) {
                const T* s = vp + p * plane + df * 4 * WIN_TOKENS;
                if (kb < 4) vf[p][df][kb] = *reinterpret_cast<const uint4*>(s + kb * 32 + g * 8);
                else { const uint2 lo = *reinterpret_cast<const uint2*>(s + 128 + g * 4); vf[p][df][kb] = make_uint4(lo.x, lo.y, 0, 0); }
            }

    const float LOG2E = 1.4426950408889634f;
    // software prefetch: Q fragment and the 9 bias/mask fragments of query block qf+1 are in flight while block qf
    // computes (one exposed L2/HBM round trip per window instead of nine)
    uint4 qn[NPL];
    uint4 bn[4];
    uint2 bn8;
#pragma unroll
    for (int p = 0; p < NPL; ++p) qn[p] = *reinterpret_cast<const uint4*>(qp + p * plane);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) bn[kb] = *reinterpret_cast<const uint4*>(bp + kb * 512 + lane * 8);
    bn8 = *reinterpret_cast<const uint2*>(bp + 2048 + lane * 4);
#pragma unroll 1
    for (int qf = 0; qf < 9; ++qf) {
        uint4 qv[NPL];
        uint4 bcur[4];
#pragma unroll
        for (int p = 0; p < NPL; ++p) qv[p] = qn[p];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) bcur[kb] = bn[kb];
        const uint2 bcur8 = bn8;
        if (qf < 8) {
#pragma unroll
            for (int p = 0; p < NPL; ++p) qn[p] = *reinterpret_cast<const uint4*>(qp + p * plane + (qf + 1) * 16 * HEAD_DIM);
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) bn[kb] = *reinterpret_cast<const uint4*>(bp + (qf + 1) * 2304 + kb * 512 + lane * 8);
            bn8 = *reinterpret_cast<const uint2*>(bp + (qf + 1) * 2304 + 2048 + lane * 4);
        }

        f32x4 s[9];
#pragma unroll
        for (int f = 0; f < 9; ++f) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            if constexpr (NPL == 2) {
                a = OpT<T>::mfma(as_v8<T>(kf[1][f]), as_v8<T>(qv[0]), a);
                a = OpT<T>::mfma(as_v8<T>(kf[0][f]), as_v8<T>(qv[1]), a);
            }
            a = OpT<T>::mfma(as_v8<T>(kf[0][f]), as_v8<T>(qv[0]), a);
            s[f] = a;
        }
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
            uint4 pf[NPL];
            split8<T, NPL>(pv, pf);
#pragma unroll
            for (int df = 0; df < 2; ++df) {
                if constexpr (NPL == 2) {
                    o[df] = OpT<T>::mfma(as_v8<T>(vf[1][df][kb]), as_v8<T>(pf[0]), o[df]);
                    o[df] = OpT<T>::mfma(as_v8<T>(vf[0][df][kb]), as_v8<T>(pf[1]), o[df]);
                }
                o[df] = OpT<T>::mfma(as_v8<T>(vf[0][df][kb]), as_v8<T>(pf[0]), o[df]);
            }
        }
        // blocked [row/16][col/32][16][32] layout: this head's 32 columns are exactly one column block
        TO* orow = out + blk_off((long long)win * WIN_TOKENS + qf * 16 + l15, head * HEAD_DIM, ld_out) + g * 8;
        const float y[8] = {o[0][0] * inv, o[0][1] * inv, o[0][2] * inv, o[0][3] * inv, o[1][0] * inv, o[1][1] * inv, o[1][2] * inv, o[1][3] * inv};
        store8_planes<TO, NPL_O>(orow, out_plane, y);
    }
}

template <class P>
hipError_t launch_attention(const AttnArgs<P>& a, hipStream_t stream) {
    constexpr int NPL_O = (P::NA > P::NW ? P::NA : P::NW);
    const long long waves = (long long)a.n_win * a.heads;
    const unsigned blocks = (unsigned)((waves + 3) / 4);
    hipLaunchKernelGGL((earth_attention_kernel<typename P::T, NPL_O>), dim3(blocks), dim3(256), 0, stream,
                       a.q, a.k, a.vt, a.plane, a.bias_exp, a.out, a.out_plane, a.ld_out, a.n_win, a.nW, a.heads);
    return hipGetLastError();
}

template hipError_t launch_attention<PrecBF16x3>(const AttnArgs<PrecBF16x3>&, hipStream_t);
template hipError_t launch_attention<PrecF16>(const AttnArgs<PrecF16>&, hipStream_t);
template hipError_t launch_attention<PrecF16x3>(const AttnArgs<PrecF16x3>&, hipStream_t);

}  // namespace skp
