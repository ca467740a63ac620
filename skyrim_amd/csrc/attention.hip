// Earth-specific 3D window attention (EarthAttention3D core) for gfx950.
//
// One wavefront = one (window, head): 144 tokens x head_dim 32.  Everything a wave needs is read
// straight into MFMA fragment registers -- Q/K rows are 64-byte contiguous, V arrives transposed
// ([d][token], written that way by the QKV epilogue) so each lane's 4 keys are 8 contiguous bytes,
// and the earth-specific bias (+ shifted-window mask) is pre-expanded per (window type, head) in
// exactly the accumulator layout.  No LDS, no barriers; the only cross-lane traffic is the 2-step
// butterfly of the softmax row reduction.
//
//   S^T[key][q] = K Q^T   (A = K frag, B = Q frag)   -> lane (q = l&15) holds keys 16f + 4(l>>4) + r
//   O^T[d][q]   = V^T P^T (A = V^T frag, B = P frag) -> lane (q = l&15) holds d = 16df + 4(l>>4) + r
//
// The P fragment for MFMA k-slot j of key block kb is S-fragment (2kb + j/4) register j%4, so the
// softmax output feeds the second MFMA without any data movement; V^T fragments use the same key order.
// NPL = 2 runs every product as hi*hi + lo*hi + hi*lo (fp32-class accuracy on the bf16 pipe).
#include "common.h"
#include "launchers.h"

namespace skp {

// Q, K, V and the softmax output P are single fp16 planes in EVERY precision mode: attention is the least
// rounding-sensitive part of the network (measured on the oracle: fp16 q/k/v -> 7e-5, fp16 P -> 4e-5 per-channel
// error, vs 3.3e-4 / 3.6e-4 for the attention output / MLP hidden, which therefore stay hi/lo split), so
// QK^T and PV are one MFMA term each and the Q/K/V round trip through HBM is 2 bytes per element.
template <class TO, int NPL_O>
__global__ void __launch_bounds__(256) earth_attention_kernel(const f16* __restrict__ q, const f16* __restrict__ k,
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
    const T* kp = k + base * (WIN_TOKENS * HEAD_DIM) + l15 * HEAD_DIM + g * 8;
    const T* vp = vt + base * (WIN_TOKENS * HEAD_DIM) + l15 * WIN_TOKENS + g * 4;
    const f16* bp = bias_exp + ((long long)type * heads + head) * (81 * 256) + lane * 4;

    uint4 kf[NPL][9];
#pragma unroll
    for (int p = 0; p < NPL; ++p)
#pragma unroll
        for (int f = 0; f < 9; ++f) kf[p][f] = *reinterpret_cast<const uint4*>(kp + p * plane + f * 16 * HEAD_DIM);

    uint4 vf[NPL][2][5];
#pragma unroll
    for (int p = 0; p < NPL; ++p)
#pragma unroll
        for (int df = 0; df < 2; ++df)
#pragma unroll
            for (int kb = 0; kb < 5; ++kb) {
                const T* s = vp + p * plane + df * 16 * WIN_TOKENS + kb * 32;
                const uint2 lo = *reinterpret_cast<const uint2*>(s);
                const uint2 hi = kb < 4 ? *reinterpret_cast<const uint2*>(s + 16) : make_uint2(0, 0);
                vf[p][df][kb] = make_uint4(lo.x, lo.y, hi.x, hi.y);
            }

    const float LOG2E = 1.4426950408889634f;
    // software prefetch: Q fragment and the 9 bias/mask fragments of query block qf+1 are in flight while block qf
    // computes (one exposed L2/HBM round trip per window instead of nine)
    uint4 qn[NPL];
    uint2 bn[9];
#pragma unroll
    for (int p = 0; p < NPL; ++p) qn[p] = *reinterpret_cast<const uint4*>(qp + p * plane);
#pragma unroll
    for (int f = 0; f < 9; ++f) bn[f] = *reinterpret_cast<const uint2*>(bp + f * 256);
#pragma unroll 1
    for (int qf = 0; qf < 9; ++qf) {
        uint4 qv[NPL];
        uint2 bcur[9];
#pragma unroll
        for (int p = 0; p < NPL; ++p) qv[p] = qn[p];
#pragma unroll
        for (int f = 0; f < 9; ++f) bcur[f] = bn[f];
        if (qf < 8) {
#pragma unroll
            for (int p = 0; p < NPL; ++p) qn[p] = *reinterpret_cast<const uint4*>(qp + p * plane + (qf + 1) * 16 * HEAD_DIM);
#pragma unroll
            for (int f = 0; f < 9; ++f) bn[f] = *reinterpret_cast<const uint2*>(bp + ((qf + 1) * 9 + f) * 256);
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
        for (int f = 0; f < 9; ++f) {
            typedef f16 h4 __attribute__((ext_vector_type(4)));
            const h4 b = __builtin_bit_cast(h4, bcur[f]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s[f][r] += (float)b[r];
                mx = fmaxf(mx, s[f][r]);
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
        TO* orow = out + blk_off((long long)win * WIN_TOKENS + qf * 16 + l15, head * HEAD_DIM, ld_out) + g * 4;
#pragma unroll
        for (int df = 0; df < 2; ++df) {
            const float y[4] = {o[df][0] * inv, o[df][1] * inv, o[df][2] * inv, o[df][3] * inv};
            store4_planes<TO, NPL_O>(orow + df * 16, out_plane, y);
        }
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
