// Tile shapes used by the Pangu stages, per precision mode.  Split modes stage hi+lo planes of both
// operands, so they use BK = 32 to stay at <= 57 KiB of LDS (2 blocks per CU); single-term modes use BK = 64.
#pragma once
#include "gemm.h"
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
};

#define SKP_CHECK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) return e_; } while (0)

}  // namespace skp
