// Attention half of an EarthSpecificBlock around the window-attention kernel:
//   op_qkv : window gather (pad + roll + partition) fused into the A loader, QKV linear, head split
//   op_proj: output linear + LayerNorm(norm1) + window reverse / un-roll / crop + residual add, in place
#include "tiles.h"

namespace skp {

template <class P>
hipError_t op_qkv(const Geom& g, const BlockW<typename P::T>& b, const int* widx, int res, const float* X, const Work<P>& wk, hipStream_t s) {
    typedef typename P::T T;
    constexpr int NPL = (P::NA > P::NW ? P::NA : P::NW);
    const int C = res == 0 ? 192 : 384, heads = C / HEAD_DIM;
    GemmArgs<P, ALRowsF32, EpQKV<T, NPL>> a;
    a.al = ALRowsF32{X, widx, C, g.mwin[res], C, 0};
    a.ep = EpQKV<T, NPL>{wk.q, wk.k, wk.vt, wk.qkv_plane, b.qkv_b, C, heads, 0.17677669529663687f};
    a.W = b.qkv.w; a.w_plane = b.qkv.plane; a.ldw = b.qkv.ldw;
    a.M = g.mwin[res]; a.N = 3 * C; a.K = C;
    return launch_gemm<P, typename Tiles<P>::G128>(a, s);
}

template <class P>
hipError_t op_proj(const Geom& g, const BlockW<typename P::T>& b, const int* widx, int res, float* X, const Work<P>& wk, hipStream_t s) {
    typedef typename ActT<P>::type S;
    typedef ALRowsAct<P, S> ALF;
    typedef EpLayerNorm<RowMapIndexed, SinkResidual> EP;
    const int C = res == 0 ? 192 : 384;
    GemmArgs<P, typename ALF::type, EP> a;
    a.al = ALF::make(wk.ao, C, g.mwin[res], C);
    a.ep = EP{RowMapIndexed{widx}, SinkResidual{X}, b.proj_b, b.n1_g, b.n1_b, 1e-5f};
    a.W = b.proj.w; a.w_plane = b.proj.plane; a.ldw = b.proj.ldw;
    a.M = g.mwin[res]; a.N = C; a.K = C;
    if (res == 0) return launch_gemm<P, typename Tiles<P>::L192>(a, s);
    return launch_gemm<P, typename Tiles<P>::L384>(a, s);
}

template hipError_t op_qkv<PrecBF16x3>(const Geom&, const BlockW<bf16>&, const int*, int, const float*, const Work<PrecBF16x3>&, hipStream_t);
template hipError_t op_qkv<PrecF16>(const Geom&, const BlockW<f16>&, const int*, int, const float*, const Work<PrecF16>&, hipStream_t);
template hipError_t op_proj<PrecBF16x3>(const Geom&, const BlockW<bf16>&, const int*, int, float*, const Work<PrecBF16x3>&, hipStream_t);
template hipError_t op_proj<PrecF16>(const Geom&, const BlockW<f16>&, const int*, int, float*, const Work<PrecF16>&, hipStream_t);

}  // namespace skp
