// Tile shapes used by the Pangu stages, per precision mode.  Split modes stage hi+lo planes of both
// operands, so they use BK = 32 to stay at <= 57 KiB of LDS (2 blocks per CU); single-term modes use BK = 64.
#pragma once
#include "gemm.h"
#include "gemm_dma.h"
#include "loaders.h"
#include "epilogues.h"
#include "launchers.h"

namespace skp {

template <class P>
struct Tiles {
    static constexpr int BKP = (P::NA == 2 || P::NW == 2) ? 32 : 64;
    typedef TileCfg<128, 128, BKP, 2, 2> G128;   // generic wide-N GEMMs (QKV, fc1, DownSample)
    typedef TileCfg<64, 192, BKP, 2, 2> L192;    // N = 192 with whole rows per block (LayerNorm epilogues, embed, recover)
    typedef TileCfg<64, 384, BKP, 2, 4> L384;    // N = 384 with whole rows per block (8 wavefronts)
    typedef TileCfg<128, 64, BKP, 4, 1> N64;     // surface PatchRecovery (N = 64)
    // DMA GEMMs: one 8-wave block per CU, wave tile 64 x 96 (72 MFMAs per 32-deep k-step in split modes)
    typedef TileCfg<256, 192, BKP, 4, 2> D192;   // N multiple of 192 (QKV, fc1, proj/fc2 at C=192, UpSample, recover)
    typedef TileCfg<128, 384, BKP, 2, 4> D384;   // N = 384 with whole rows per block (proj/fc2 at C=384)
};

template <class P> constexpr int planes_of() { return P::NA > P::NW ? P::NA : P::NW; }

#define SKP_CHECK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return e_; } while (0)

}  // namespace skp
