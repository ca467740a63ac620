// PatchEmbedding (Conv3d/Conv2d with stride = kernel as im2col GEMMs, fp32 source -> register-staged GEMM)
// and PatchRecovery (ConvTranspose3d/2d as DMA GEMMs over concat(skip, x) with a scatter + crop +
// de-normalise epilogue).
#include "tiles.h"

namespace skp {

template <class P>
hipError_t op_embed(const Geom& g, const ModelW<typename P::T>& w, const float* state, typename P::T* X1s, const Work<P>& wk, hipStream_t s) {
    typedef typename P::T T;
    typedef typename Tiles<P>::L192 TC;
    typedef EpStorePlanes<T> EP;
    const int hw = g.H1 * g.W1;
    {   // surface slab -> token level 0
        GemmArgs<P, ALIm2colSurface, EP> a;
        a.al = ALIm2colSurface{state, w.masks, w.mean, w.istd, g.n_lat, g.n_lon, g.lat_top, g.H1, g.W1, g.surf0, hw};
        a.ep = EP{w.embed_s_b, 192, 0, X1s, wk.xs_plane[0]};
        a.W = w.embed_s.w; a.w_plane = w.embed_s.plane; a.ldw = w.embed_s.ldw;
        a.M = hw; a.N = 192; a.K = 128;
        SKP_CHECK((launch_gemm<P, TC>(a, s)));
    }
    {   // upper air -> token levels 1..7
        GemmArgs<P, ALIm2colUpper, EP> a;
        a.al = ALIm2colUpper{state, w.mean, w.istd, g.n_lat, g.n_lon, g.lat_top, g.H1, g.W1, g.n_levels, (g.Z - 1) * hw};
        a.ep = EP{w.embed_u_b, 192, hw, X1s, wk.xs_plane[0]};
        a.W = w.embed_u.w; a.w_plane = w.embed_u.plane; a.ldw = w.embed_u.ldw;
        a.M = (g.Z - 1) * hw; a.N = 192; a.K = 160;
        SKP_CHECK((launch_gemm<P, TC>(a, s)));
    }
    return hipSuccess;
}

template <class P>
hipError_t op_recover(const Geom& g, const ModelW<typename P::T>& w, const typename P::T* skip_s, const typename P::T* x4_s, float* state,
                      const Work<P>& wk, hipStream_t s) {
    typedef typename P::T T;
    const int hw = g.H1 * g.W1;
    {
        DmaArgs<P, AConcatPlanes<T>, EpRecover> a;
        a.as = AConcatPlanes<T>{skip_s, x4_s, wk.xs_plane[0], 192, hw, 0};
        a.ep = EpRecover{state, w.rec_s_b, w.mean, w.std, g.n_lat, g.n_lon, g.lat_top, g.H1, g.W1, g.n_levels, 1, g.surf0};
        a.W = w.rec_s.w; a.w_plane = w.rec_s.plane; a.ldw = w.rec_s.ldw; a.zrow = wk.zrow;
        a.M = hw; a.N = 64; a.K = 384;
        SKP_CHECK((launch_gemm_dma<P, typename Tiles<P>::N64>(a, s)));
    }
    {
        DmaArgs<P, AConcatPlanes<T>, EpRecover> a;
        a.as = AConcatPlanes<T>{skip_s, x4_s, wk.xs_plane[0], 192, (g.Z - 1) * hw, hw};
        a.ep = EpRecover{state, w.rec_u_b, w.mean, w.std, g.n_lat, g.n_lon, g.lat_top, g.H1, g.W1, g.n_levels, 0, g.surf0};
        a.W = w.rec_u.w; a.w_plane = w.rec_u.plane; a.ldw = w.rec_u.ldw; a.zrow = wk.zrow;
        a.M = (g.Z - 1) * hw; a.N = 160; a.K = 384;
        SKP_CHECK((launch_gemm_dma<P, typename Tiles<P>::D192>(a, s)));
    }
    return hipSuccess;
}

template hipError_t op_embed<PrecBF16x3>(const Geom&, const ModelW<bf16>&, const float*, bf16*, const Work<PrecBF16x3>&, hipStream_t);
template hipError_t op_embed<PrecF16x3>(const Geom&, const ModelW<f16>&, const float*, f16*, const Work<PrecF16x3>&, hipStream_t);
template hipError_t op_recover<PrecBF16x3>(const Geom&, const ModelW<bf16>&, const bf16*, const bf16*, float*, const Work<PrecBF16x3>&, hipStream_t);
template hipError_t op_recover<PrecF16x3>(const Geom&, const ModelW<f16>&, const f16*, const f16*, float*, const Work<PrecF16x3>&, hipStream_t);

}  // namespace skp
