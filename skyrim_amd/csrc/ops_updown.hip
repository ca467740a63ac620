// DownSample (2x2 merge + LayerNorm(768) prologue over the fp32 stream + Linear 768->384) and UpSample
// (Linear 384->768 + pixel shuffle + crop + LayerNorm(192) epilogue, then Linear 192->192).
#include "tiles.h"

namespace skp {

template <class P>
hipError_t op_down(const Geom& g, const ModelW<typename P::T>& w, const typename P::T* X1s, typename P::T* X2s, const Work<P>& wk, hipStream_t s) {
    typedef typename P::T T;
    typedef EpStorePlanes<T> EP;
    SKP_CHECK((merge_stats<T>(X1s, wk.xs_plane[0], wk.stats, g.Z, g.H1, g.W1, g.H2, g.W2, 192, 1e-5f, s)));
    GemmArgs<P, ALMergeLN<T>, EP> a;
    a.al = ALMergeLN<T>{X1s, wk.xs_plane[0], wk.stats, w.down_g, w.down_b, g.H1, g.W1, g.H2, g.W2, 192, g.ntok[1]};
    a.ep = EP{nullptr, 384, 0, X2s, wk.xs_plane[1]};
    a.W = w.down.w; a.w_plane = w.down.plane; a.ldw = w.down.ldw;
    a.M = g.ntok[1]; a.N = 384; a.K = 768;
    // whole N per block: the merged + normalised A rows are produced once; 128 of them per block: a k-step's 48 KiB weight tile (hi + lo) comes
    // from L2 once per 128 rows -- with 64-row blocks that traffic (2.3 GB per launch) was the bound: 0.45 -> 0.33 ms (docs/experiments.md A6.6)
    return launch_gemm<P, typename Tiles<P>::D384>(a, s);
}

template <class P>
hipError_t op_up(const Geom& g, const ModelW<typename P::T>& w, const typename P::T* X2s, typename P::T* X4s, const Work<P>& wk, hipStream_t s) {
    typedef typename P::T T;
    typedef typename Tiles<P>::D192 TC;
    {
        typedef EpLayerNorm<RowMapPixelShuffle, SinkStore<T, P::NA>> EP;
        DmaArgs<P, APlanes<T>, EP> a;
        a.as = APlanes<T>{X2s, wk.xs_plane[1], 384, nullptr, g.ntok[1]};
        a.ep = EP{RowMapPixelShuffle{g.H1, g.W1, g.H2, g.W2}, SinkStore<T, P::NA>{wk.u, wk.u_plane}, nullptr, w.up_g, w.up_b, 1e-5f};
        a.W = w.up1.w; a.w_plane = w.up1.plane; a.ldw = w.up1.ldw; a.zrow = wk.zrow;
        a.M = g.ntok[1]; a.N = 768; a.K = 384;
        SKP_CHECK((launch_gemm_dma<P, TC>(a, s)));
    }
    {
        typedef EpStorePlanes<T> EP;
        DmaArgs<P, APlanes<T>, EP> a;
        a.as = APlanes<T>{wk.u, wk.u_plane, 192, nullptr, g.ntok[0]};
        a.ep = EP{nullptr, 192, 0, X4s, wk.xs_plane[0]};
        a.W = w.up2.w; a.w_plane = w.up2.plane; a.ldw = w.up2.ldw; a.zrow = wk.zrow;
        a.M = g.ntok[0]; a.N = 192; a.K = 192;
        SKP_CHECK((launch_gemm_dma<P, TC>(a, s)));
    }
    return hipSuccess;
}

template hipError_t op_down<PrecBF16x3>(const Geom&, const ModelW<bf16>&, const bf16*, bf16*, const Work<PrecBF16x3>&, hipStream_t);
template hipError_t op_down<PrecF16x3>(const Geom&, const ModelW<f16>&, const f16*, f16*, const Work<PrecF16x3>&, hipStream_t);
template hipError_t op_up<PrecBF16x3>(const Geom&, const ModelW<bf16>&, const bf16*, bf16*, const Work<PrecBF16x3>&, hipStream_t);
template hipError_t op_up<PrecF16x3>(const Geom&, const ModelW<f16>&, const f16*, f16*, const Work<PrecF16x3>&, hipStream_t);

}  // namespace skp
