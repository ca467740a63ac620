// MLP half of an EarthSpecificBlock:  x += LayerNorm(norm2)( fc2( GELU( fc1(x) ) ) ), in place.
#include "tiles.h"

namespace skp {

template <class P>
hipError_t op_fc1(const Geom& g, const BlockW<typename P::T>& b, int res, const float* X, const Work<P>& wk, hipStream_t s) {
    typedef typename ActT<P>::type S;
    const int C = res == 0 ? 192 : 384;
    GemmArgs<P, ALRowsF32, EpGelu<S>> a;
    a.al = ALRowsF32{X, nullptr, C, g.ntok[res], C, 0};
    a.ep = EpGelu<S>{wk.hid, b.fc1_b, 4 * C};
    a.W = b.fc1.w; a.w_plane = b.fc1.plane; a.ldw = b.fc1.ldw;
    a.M = g.ntok[res]; a.N = 4 * C; a.K = C;
    return launch_gemm<P, typename Tiles<P>::G128>(a, s);
}

template <class P>
hipError_t op_fc2(const Geom& g, const BlockW<typename P::T>& b, int res, float* X, const Work<P>& wk, hipStream_t s) {
    typedef typename ActT<P>::type S;
    typedef ALRowsAct<P, S> ALF;
    typedef EpLayerNorm<RowMapIndexed, SinkResidual> EP;
    const int C = res == 0 ? 192 : 384;
    GemmArgs<P, typename ALF::type, EP> a;
    a.al = ALF::make(wk.hid, 4 * C, g.ntok[res], 4 * C);
    a.ep = EP{RowMapIndexed{nullptr}, SinkResidual{X}, b.fc2_b, b.n2_g, b.n2_b, 1e-5f};
    a.W = b.fc2.w; a.w_plane = b.fc2.plane; a.ldw = b.fc2.ldw;
    a.M = g.ntok[res]; a.N = C; a.K = 4 * C;
    if (res == 0) return launch_gemm<P, typename Tiles<P>::L192>(a, s);
    return launch_gemm<P, typename Tiles<P>::L384>(a, s);
}

template hipError_t op_fc1<PrecBF16x3>(const Geom&, const BlockW<bf16>&, int, const float*, const Work<PrecBF16x3>&, hipStream_t);
template hipError_t op_fc1<PrecF16>(const Geom&, const BlockW<f16>&, int, const float*, const Work<PrecF16>&, hipStream_t);
template hipError_t op_fc2<PrecBF16x3>(const Geom&, const BlockW<bf16>&, int, float*, const Work<PrecBF16x3>&, hipStream_t);
template hipError_t op_fc2<PrecF16>(const Geom&, const BlockW<f16>&, int, float*, const Work<PrecF16>&, hipStream_t);

}  // namespace skp
