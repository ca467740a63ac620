// MLP half of an EarthSpecificBlock:  x += LayerNorm(norm2)( fc2( GELU( fc1(x) ) ) ), in place.
#include <type_traits>
#include "tiles.h"

namespace skp {

template <class P>
hipError_t op_fc1(const Geom& g, const BlockW<typename P::T>& b, int res, const typename P::T* Xs, const Work<P>& wk, hipStream_t s) {
    typedef typename P::T T;
    typedef EpGelu<T, P::NA> EP;
    const int C = res == 0 ? 192 : 384;
    if constexpr (P::NA == 2) {
        if (wk.hid16) {   // hidden as one fp16 plane: half the HBM bytes of the largest intermediate
            typedef EpGelu<f16, 1> EPH;
            DmaArgs<P, APlanes<T>, EPH> a;
            a.as = APlanes<T>{Xs, wk.xs_plane[res], C, nullptr, g.ntok[res]};
            a.ep = EPH{reinterpret_cast<f16*>(wk.hid), 0, b.fc1_b, 4 * C};
            a.W = b.fc1.w; a.w_plane = b.fc1.plane; a.ldw = b.fc1.ldw; a.zrow = wk.zrow;
            a.M = g.ntok[res]; a.N = 4 * C; a.K = C;
            return launch_gemm_dma<P, typename Tiles<P>::D192>(a, s);
        }
    }
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
    if constexpr (P::NA == 2) {
        if (wk.hid16) {
            typedef PrecF16x2W P2;
            DmaArgs<P2, APlanes<f16>, EP> a;
            a.as = APlanes<f16>{reinterpret_cast<const f16*>(wk.hid), 0, 4 * C, nullptr, g.ntok[res]};
            a.ep = EP{RowMapIndexed{nullptr}, SinkResidual<T>{Xs, wk.xs_plane[res]}, b.fc2_b, b.n2_g, b.n2_b, 1e-5f};
            // fp16 hi/lo weights: the engine's own fc2 planes when T is fp16, the extra fc2h copy in the bf16 engine
            if constexpr (std::is_same<T, f16>::value) { a.W = b.fc2.w; a.w_plane = b.fc2.plane; a.ldw = b.fc2.ldw; }
            else { a.W = b.fc2h.w; a.w_plane = b.fc2h.plane; a.ldw = b.fc2h.ldw; }
            a.zrow = reinterpret_cast<const f16*>(wk.zrow);
            a.M = g.ntok[res]; a.N = C; a.K = 4 * C;
            if (res == 0) return launch_gemm_dma<P2, typename Tiles<P2>::D192>(a, s);
            return launch_gemm_dma<P2, typename Tiles<P2>::D384>(a, s);
        }
    }
    DmaArgs<P, APlanes<T>, EP> a;
    a.as = APlanes<T>{wk.hid, wk.hid_plane, 4 * C, nullptr, g.ntok[res]};
    a.ep = EP{RowMapIndexed{nullptr}, SinkResidual<T>{Xs, wk.xs_plane[res]}, b.fc2_b, b.n2_g, b.n2_b, 1e-5f};
    a.W = b.fc2.w; a.w_plane = b.fc2.plane; a.ldw = b.fc2.ldw; a.zrow = wk.zrow;
    a.M = g.ntok[res]; a.N = C; a.K = 4 * C;
    if (res == 0) return launch_gemm_dma<P, typename Tiles<P>::D192>(a, s);
    return launch_gemm_dma<P, typename Tiles<P>::D384>(a, s);
}

template hipError_t op_fc1<PrecBF16x3>(const Geom&, const BlockW<bf16>&, int, const bf16*, const Work<PrecBF16x3>&, hipStream_t);
template hipError_t op_fc1<PrecF16>(const Geom&, const BlockW<f16>&, int, const f16*, const Work<PrecF16>&, hipStream_t);
template hipError_t op_fc1<PrecF16x3>(const Geom&, const BlockW<f16>&, int, const f16*, const Work<PrecF16x3>&, hipStream_t);
template hipError_t op_fc2<PrecBF16x3>(const Geom&, const BlockW<bf16>&, int, bf16*, const Work<PrecBF16x3>&, hipStream_t);
template hipError_t op_fc2<PrecF16>(const Geom&, const BlockW<f16>&, int, f16*, const Work<PrecF16>&, hipStream_t);
template hipError_t op_fc2<PrecF16x3>(const Geom&, const BlockW<f16>&, int, f16*, const Work<PrecF16x3>&, hipStream_t);

}  // namespace skp
