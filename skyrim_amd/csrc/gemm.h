// LDS-tiled MFMA GEMM for gfx950 with pluggable A-operand loaders and fused epilogues.
//
//   out = epilogue( A[M,K] * W[N,K]^T )
//
// A rows are produced by a *loader* (plain rows, window gather, im2col, 2x2 merge + LN, concat)
// as fp32 values that are rounded/split to the 16-bit MFMA operand type on their way into LDS;
// W is a pre-split 16-bit [N][ldw] matrix (hi plane, optional lo plane).  The product is formed
// by 1..3 MFMA terms per tile (common.h: precision modes) with fp32 accumulation.
//
// Tile: BM x BN x BK, WM x WN wavefronts of 64 lanes, each owning a (BM/WM) x (BN/WN) sub-tile as
// FM x FN fragments of v_mfma_f32_16x16x32.  The default operand order is "swapped"
// (D^T = W A^T) so that a lane ends up with 4 consecutive n for one m -> 16-byte row-major stores
// and row-wise (LayerNorm) reductions that stay inside 4 lanes + registers.
//
// Pipeline: register prefetch of k-tile t+1 (global -> VGPR) overlaps the MFMAs of tile t; one LDS
// buffer, two barriers per k-tile; several blocks per CU hide the rest.
#pragma once
#include "common.h"
#include "epilogues.h"

namespace skp {

template <int BM_, int BN_, int BK_, int WM_, int WN_>
struct TileCfg {
    static constexpr int BM = BM_, BN = BN_, BK = BK_, WM = WM_, WN = WN_;
    static constexpr int THREADS = 64 * WM * WN;
    static constexpr int WTM = BM / WM, WTN = BN / WN;
    static constexpr int FM = WTM / 16, FN = WTN / 16;
    static constexpr int CPR = BK / 8;                                   // 16-byte chunks per tile row
    static constexpr int A_CHUNKS = (BM * CPR + THREADS - 1) / THREADS;  // per thread
    static constexpr int W_CHUNKS = (BN * CPR + THREADS - 1) / THREADS;
    static_assert(WTM % 16 == 0 && WTN % 16 == 0, "wave tile must be a multiple of 16x16");
    static_assert(THREADS % CPR == 0, "thread->slot mapping");
    static_assert(BK == 32 || BK == 64, "BK");
};

template <class P, class AL, class EP>
struct GemmArgs {
    AL al;
    EP ep;
    const typename P::T* W;   // [N][ldw] hi plane
    long long w_plane;        // element offset of the lo plane (NW == 2)
    int ldw;
    int M, N, K;              // K: loop bound (multiple of 8; loader and W both zero-fill beyond their own extent)
};

// loaders that take the k-tile (wave-uniform) and the thread's slot separately declare  static constexpr bool kUniformK = true
template <class AL, class = void> struct al_uniform_k_t { static constexpr bool value = false; };
template <class AL> struct al_uniform_k_t<AL, decltype((void)AL::kUniformK)> { static constexpr bool value = AL::kUniformK; };
template <class AL> constexpr bool al_uniform_k_v = al_uniform_k_t<AL>::value;
// loaders whose ROWS are the contiguous index in memory (k strided) declare  static constexpr bool kRowsFirst = true : the staging threads
// are then laid out rows-first (a wave = 64 consecutive rows of ONE 8-wide k chunk), so that each of a thread's 8 scalar loads is, across the
// wave, one contiguous 256-byte segment; with the default k-first layout (4 k chunks x 16 rows per wave) the same load touches four
// 64-byte segments of four different k rows -- half cache lines, four times the requests
template <class AL, class = void> struct al_rows_first_t { static constexpr bool value = false; };
template <class AL> struct al_rows_first_t<AL, decltype((void)AL::kRowsFirst)> { static constexpr bool value = AL::kRowsFirst; };
template <class AL> constexpr bool al_rows_first_v = al_rows_first_t<AL>::value;

template <class P, class TC>
constexpr int gemm_smem_bytes() { return (P::NA * TC::BM + P::NW * TC::BN) * TC::BK * 2; }

template <class P, class TC, class AL, class EP, bool SWAP>
__device__ __forceinline__ void gemm_body(const GemmArgs<P, AL, EP>& g, char* smem, const int tile_x = blockIdx.x, const int tile_y = blockIdx.y) {
    typedef typename P::T T;
    constexpr int NA = P::NA, NW = P::NW;
    constexpr int BM = TC::BM, BN = TC::BN, BK = TC::BK, CPR = TC::CPR, THREADS = TC::THREADS;
    constexpr int FM = TC::FM, FN = TC::FN;
    constexpr int A_PLANE = BM * BK * 2, W_PLANE = BN * BK * 2;
    constexpr int ROWS_PER_PASS = THREADS / CPR;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / TC::WN, wn = wave % TC::WN;
    const int m0 = tile_y * BM, n0 = tile_x * BN;          // (the launch's own block indices unless the caller re-maps them: sfno_ops.hip)
    char* As = smem;
    char* Ws = smem + NA * A_PLANE;

    const int st_slot = tid % CPR;          // this thread's 16-byte slot within a tile row
    const int st_row = tid / CPR;           // first tile row it stages
    // the A operand's own thread layout (rows-first loaders, one chunk per thread: see al_rows_first_t); W keeps the k-first one
    bool rows_first = false;                  // (wave-uniform; the loader decides from its strides: ALFast<false> always, ALStrided when k is strided)
    if constexpr (al_rows_first_v<AL> && TC::A_CHUNKS == 1 && THREADS == BM * CPR) rows_first = g.al.rows_first();
    const int a_slot = rows_first ? tid / BM : st_slot;
    const int a_row = rows_first ? tid % BM : st_row;

    typename AL::Row arow[TC::A_CHUNKS];
#pragma unroll
    for (int i = 0; i < TC::A_CHUNKS; ++i) {
        const int r = a_row + i * ROWS_PER_PASS;
        arow[i] = g.al.row(m0 + (r < BM ? r : 0));
    }
    const T* wrow[TC::W_CHUNKS];
#pragma unroll
    for (int j = 0; j < TC::W_CHUNKS; ++j) {
        const int r = st_row + j * ROWS_PER_PASS;
        const int n = n0 + r;
        wrow[j] = (r < BN && n < g.N) ? g.W + (long long)n * g.ldw : nullptr;
    }

    f32x4 acc[FM][FN];
#pragma unroll
    for (int a = 0; a < FM; ++a)
#pragma unroll
        for (int b = 0; b < FN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    typename AL::Raw araw[TC::A_CHUNKS];
    uint4 wraw[NW][TC::W_CHUNKS];

    auto load_tile = [&](int kt) {
        const int k = kt * BK + st_slot * 8;
#pragma unroll
        for (int i = 0; i < TC::A_CHUNKS; ++i) {
            if (TC::A_CHUNKS * ROWS_PER_PASS == BM || a_row + i * ROWS_PER_PASS < BM) {
                if constexpr (al_uniform_k_v<AL>) g.al.issue2(arow[i], kt * BK, a_slot * 8, BK, araw[i]);     // uniform base + per-thread offset
                else g.al.issue(arow[i], kt * BK + a_slot * 8, araw[i]);
            }
        }
#pragma unroll
        for (int j = 0; j < TC::W_CHUNKS; ++j) {
            const bool ok = wrow[j] != nullptr && k < g.K;
#pragma unroll
            for (int p = 0; p < NW; ++p)
                wraw[p][j] = ok ? *reinterpret_cast<const uint4*>(wrow[j] + p * g.w_plane + k) : make_uint4(0, 0, 0, 0);
        }
    };
    auto stage_tile = [&]() {
#pragma unroll
        for (int i = 0; i < TC::A_CHUNKS; ++i) {
            const int r = a_row + i * ROWS_PER_PASS;
            if (TC::A_CHUNKS * ROWS_PER_PASS == BM || r < BM) {
                uint4 o[NA];
                if constexpr (AL::kDirect) {
                    o[0] = g.al.direct(araw[i]);
                } else {
                    float v[8];
                    g.al.finish(araw[i], v);
                    split8<T, NA>(v, o);
                }
#pragma unroll
                for (int p = 0; p < NA; ++p) *reinterpret_cast<uint4*>(As + p * A_PLANE + lds_off<BK>(r, a_slot)) = o[p];
            }
        }
#pragma unroll
        for (int j = 0; j < TC::W_CHUNKS; ++j) {
            const int r = st_row + j * ROWS_PER_PASS;
            if (TC::W_CHUNKS * ROWS_PER_PASS == BN || r < BN) {
#pragma unroll
                for (int p = 0; p < NW; ++p) *reinterpret_cast<uint4*>(Ws + p * W_PLANE + lds_off<BK>(r, st_slot)) = wraw[p][j];
            }
        }
    };

    const int nk = (g.K + BK - 1) / BK;
    const int fr_row = lane & 15, fr_grp = lane >> 4;
    load_tile(0);
    g.ep.template init<TC>(smem + gemm_smem_bytes<P, TC>() + kEpiReduceBytes, tid, n0);   // ordered by the loop's first barrier
    for (int kt = 0; kt < nk; ++kt) {
        stage_tile();
        __syncthreads();
        if (kt + 1 < nk) load_tile(kt + 1);
#pragma unroll
        for (int ks = 0; ks < BK / 32; ++ks) {
            const int slot = ks * 4 + fr_grp;
            uint4 af[NA][FM];
#pragma unroll
            for (int a = 0; a < FM; ++a) {
                const int r = wm * TC::WTM + a * 16 + fr_row;
#pragma unroll
                for (int p = 0; p < NA; ++p) af[p][a] = *reinterpret_cast<const uint4*>(As + p * A_PLANE + lds_off<BK>(r, slot));
            }
#pragma unroll
            for (int b = 0; b < FN; ++b) {
                const int r = wn * TC::WTN + b * 16 + fr_row;
                uint4 wf[NW];
#pragma unroll
                for (int p = 0; p < NW; ++p) wf[p] = *reinterpret_cast<const uint4*>(Ws + p * W_PLANE + lds_off<BK>(r, slot));
                // small terms first, the hi*hi term last
                if constexpr (NW == 2) {
#pragma unroll
                    for (int a = 0; a < FM; ++a)
                        acc[a][b] = SWAP ? OpT<T>::mfma(as_v8<T>(wf[1]), as_v8<T>(af[0][a]), acc[a][b])
                                         : OpT<T>::mfma(as_v8<T>(af[0][a]), as_v8<T>(wf[1]), acc[a][b]);
                }
                if constexpr (NA == 2) {
#pragma unroll
                    for (int a = 0; a < FM; ++a)
                        acc[a][b] = SWAP ? OpT<T>::mfma(as_v8<T>(wf[0]), as_v8<T>(af[1][a]), acc[a][b])
                                         : OpT<T>::mfma(as_v8<T>(af[1][a]), as_v8<T>(wf[0]), acc[a][b]);
                }
#pragma unroll
                for (int a = 0; a < FM; ++a)
                    acc[a][b] = SWAP ? OpT<T>::mfma(as_v8<T>(wf[0]), as_v8<T>(af[0][a]), acc[a][b])
                                     : OpT<T>::mfma(as_v8<T>(af[0][a]), as_v8<T>(wf[0]), acc[a][b]);
            }
        }
        __syncthreads();
    }
    g.ep.template run<TC, SWAP>(acc, m0 + wm * TC::WTM, n0 + wn * TC::WTN, lane, wm, wn, smem + gemm_smem_bytes<P, TC>(), g.M, g.N, tile_x);
}

template <class P, class TC, class AL, class EP>
__global__ void __launch_bounds__(TC::THREADS) gemm_kernel(const GemmArgs<P, AL, EP> g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if constexpr (EP::kDualOrder) {
        if (g.ep.unswapped(blockIdx.x * TC::BN)) {
            gemm_body<P, TC, AL, EP, false>(g, smem);
            return;
        }
    }
    gemm_body<P, TC, AL, EP, true>(g, smem);
}

template <class P, class TC, class AL, class EP>
inline hipError_t launch_gemm(const GemmArgs<P, AL, EP>& g, hipStream_t stream) {
    dim3 grid((g.N + TC::BN - 1) / TC::BN, (g.M + TC::BM - 1) / TC::BM);
    if (grid.x == 0 || grid.y == 0) return hipSuccess;
    constexpr int smem = gemm_smem_bytes<P, TC>() + kEpiScratch;
    static_assert(smem <= 160 * 1024, "LDS per block");
    if (smem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_kernel<P, TC, AL, EP>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((gemm_kernel<P, TC, AL, EP>), grid, dim3(TC::THREADS), smem, stream, g);
    return hipGetLastError();
}

// element (a, b, r) of a wave's accumulator tile, swapped order:   m = m0w + a*16 + (lane&15),
//                                                                  n = n0w + b*16 + 4*(lane>>4) + r
//                                       un-swapped order:          n = n0w + b*16 + (lane&15),
//                                                                  m = m0w + a*16 + 4*(lane>>4) + r

}  // namespace skp
