// Attention half of an EarthSpecificBlock around the window-attention kernel:
//   op_qkv : window gather (pad + roll + partition) as per-row DMA sources, QKV linear, head split
//   op_proj: output linear + LayerNorm(norm1) + window reverse / un-roll / crop + residual add, in place
#include <type_traits>
#include "tiles.h"

namespace skp {

template <class P>
hipError_t op_qkv(const Geom& g, const BlockW<typename P::T>& b, const int* widx, int res, const typename P::T* Xs, const Work<P>& wk, hipStream_t s) {
    typedef typename P::T T;
    const int C = res == 0 ? 192 : 384, heads = C / HEAD_DIM;
    if constexpr (std::is_same<P, PrecF16x3>::value) {
        if (wk.qkv_a1) {   // Q/K/V tolerate an 11-bit A operand (tools/sensitivity.py: 6.6e-5): hi plane only, 2 terms
            typedef PrecF16x2W P2;
            DmaArgs<P2, APlanes<f16>, EpQKV<f16, 1>> a;
            a.as = APlanes<f16>{Xs, 0, C, widx, g.mwin[res]};
            a.ep = EpQKV<f16, 1>{wk.q, wk.k, wk.vt, wk.qkv_plane, b.qkv_b, C, heads, 0.17677669529663687f};
            a.W = b.qkv.w; a.w_plane = b.qkv.plane; a.ldw = b.qkv.ldw; a.zrow = wk.zrow;
            a.M = g.mwin[res]; a.N = 3 * C; a.K = C;
            return launch_gemm_dma<P2, typename Tiles<P2>::D192>(a, s);
        }
    }
    DmaArgs<P, APlanes<T>, EpQKV<f16, 1>> a;
    a.as = APlanes<T>{Xs, wk.xs_plane[res], C, widx, g.mwin[res]};
    a.ep = EpQKV<f16, 1>{wk.q, wk.k, wk.vt, wk.qkv_plane, b.qkv_b, C, heads, 0.17677669529663687f};
    a.W = b.qkv.w; a.w_plane = b.qkv.plane; a.ldw = b.qkv.ldw; a.zrow = wk.zrow;
    a.M = g.mwin[res]; a.N = 3 * C; a.K = C;
    return launch_gemm_dma<P, typename Tiles<P>::D192>(a, s);
}

template <class P>
hipError_t op_proj(const Geom& g, const BlockW<typename P::T>& b, const int* widx, int res, typename P::T* Xs, const Work<P>& wk, hipStream_t s) {
    typedef typename P::T T;
    typedef EpLayerNorm<RowMapIndexed, SinkResidual<T>> EP;
    const int C = res == 0 ? 192 : 384;
    DmaArgs<P, APlanes<T>, EP> a;
    a.as = APlanes<T>{wk.ao, wk.ao_plane, C, nullptr, g.mwin[res]};
    a.ep = EP{RowMapIndexed{widx}, SinkResidual<T>{Xs, wk.xs_plane[res]}, b.proj_b, b.n1_g, b.n1_b, 1e-5f};
    a.W = b.proj.w; a.w_plane = b.proj.plane; a.ldw = b.proj.ldw; a.zrow = wk.zrow;
    a.M = g.mwin[res]; a.N = C; a.K = C;
    if (res == 0) return launch_gemm_dma<P, typename Tiles<P>::D192>(a, s);
    return launch_gemm_dma<P, typename Tiles<P>::D384>(a, s);
}

template hipError_t op_qkv<PrecBF16x3>(const Geom&, const BlockW<bf16>&, const int*, int, const bf16*, const Work<PrecBF16x3>&, hipStream_t);
template hipError_t op_qkv<PrecF16x3>(const Geom&, const BlockW<f16>&, const int*, int, const f16*, const Work<PrecF16x3>&, hipStream_t);
template hipError_t op_proj<PrecBF16x3>(const Geom&, const BlockW<bf16>&, const int*, int, bf16*, const Work<PrecBF16x3>&, hipStream_t);
template hipError_t op_proj<PrecF16x3>(const Geom&, const BlockW<f16>&, const int*, int, f16*, const Work<PrecF16x3>&, hipStream_t);

}  // namespace skp
