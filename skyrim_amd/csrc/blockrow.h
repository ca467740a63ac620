// The argument block of the one-plane-weight block kernel (fused_block2.hip) and the register ring through which its weight fragments leave
// LDS.  gfx950 only.
#pragma once
#include "gemm_dma.h"

namespace skp {

template <class T>
struct Block2Args {
    const T* ao; long long ao_plane;       // attention output, window-ordered rows, blocked layout, hi / lo planes
    int M;                                 // stream tokens (multiple of 16)
    T* xs; long long xs_plane;             // residual stream planes, blocked layout
    const int* winv;                       // stream token -> window row of the attention output
    const T *projh, *w1h, *w2h;            // fragment-order weights, hi plane only (prep_rowtile_weights / prep_mlp_weights with planes = 1)
    const float *proj_b, *g1, *e1, *b1, *b2, *g2, *e2;
    float eps;
};

// two consecutive 1 KiB fragments; the scheduling barrier pins the reads HERE in program order (fused_mlp.hip: ld_pair)
__device__ __forceinline__ void sk_ld2(const char* p, uint4 (&w)[2]) {
    w[0] = *reinterpret_cast<const uint4*>(p);
    w[1] = *reinterpret_cast<const uint4*>(p + 1024);
    __builtin_amdgcn_sched_barrier(0);
}

// NP fragment pairs at consecutive KiB of `st`, through a ring of RD register pairs read RD - 1 pairs ahead of their MFMAs
template <int NP, int RD, class Body>
__device__ __forceinline__ void sk_stream(const char* st, Body&& body) {
    uint4 ring[RD][2];
#pragma unroll
    for (int p = 0; p < RD - 1 && p < NP; ++p) sk_ld2(st + (p << 11), ring[p % RD]);
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        if (p + RD - 1 < NP) sk_ld2(st + ((p + RD - 1) << 11), ring[(p + RD - 1) % RD]);
        body(p, ring[p % RD][0], ring[p % RD][1]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

}  // namespace skp
