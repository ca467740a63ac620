// MLP half of an EarthSpecificBlock:  x += LayerNorm(norm2)( fc2( GELU( fc1(x) ) ) ), in place.
#include <type_traits>
#include "tiles.h"

namespace skp {

template <class P>
hipError_t op_fc1(const Geom& g, const BlockW<typename P::T>& b, int res, const typename P::T* Xs, const Work<P>& wk, hipStream_t s) {
    typedef typename P::T T;
    typedef EpGelu<T, P::NA> EP;
    const int C = res == 0 ? 192 : 384;
    DmaArgs<P, APlanes<T>, EP> a;
    a.as = APlanes<T>{Xs, wk.xs_plane[res], C, nullptr, g.ntok[res]};
    a.ep = EP{wk.hid, wk.hid_plane, b.fc1_b, 4 * C};
    a.W = b.fc1.w; a.w_plane = b.fc1.plane; a.ldw = b.fc1.ldw; a.zrow = wk.zrow;
    a.M = g.ntok[res]; a.N = 4 * C; a.K = C;
    return launch_gemm_dma<P, typename Tiles<P>::D192>(a, s);
}

template <class P>
hipError_t op_fc2(const Geom& g, const BlockW<typename P::T>& b, int res, typename P::T* Xs, const Work<P>& wk, hipStream_t s) {
    typedef typename P::T T;
    typedef EpLayerNorm<RowMapIndexed, SinkResidual<T>> EP;
    const int C = res == 0 ? 192 : 384;
    DmaArgs<P, APlanes<T>, EP> a;
    a.as = APlanes<T>{wk.hid, wk.hid_plane, 4 * C, nullptr, g.ntok[res]};
    a.ep = EP{RowMapIndexed{nullptr}, SinkResidual<T>{Xs, wk.xs_plane[res]}, b.fc2_b, b.n2_g, b.n2_b, 1e-5f};
    a.W = b.fc2.w; a.w_plane = b.fc2.plane; a.ldw = b.fc2.ldw; a.zrow = wk.zrow;
    a.M = g.ntok[res]; a.N = C; a.K = 4 * C;
    if (res == 0) return launch_gemm_dma<P, typename Tiles<P>::D192>(a, s);
    return launch_gemm_dma<P, typename Tiles<P>::D384>(a, s);
}

template hipError_t op_fc1<PrecBF16x3>(const Geom&, const BlockW<bf16>&, int, const bf16*, const Work<PrecBF16x3>&, hipStream_t);
template hipError_t op_fc1<PrecF16x3>(const Geom&, const BlockW<f16>&, int, const f16*, const Work<PrecF16x3>&, hipStream_t);
template hipError_t op_fc2<PrecBF16x3>(const Geom&, const BlockW<bf16>&, int, bf16*, const Work<PrecBF16x3>&, hipStream_t);
template hipError_t op_fc2<PrecF16x3>(const Geom&, const BlockW<f16>&, int, f16*, const Work<PrecF16x3>&, hipStream_t);

}  // namespace skp
