// "Ping-pong" LDS-DMA GEMM for the store-heavy stages (fc1, QKV): a CU can push only ~8 B/clk of stores towards L2
// (tools/micro/store_burst.hip: 18.8 GB/s per CU whatever the store width or the number of CUs storing), so an epilogue
// that writes a 256x192 tile (196 KB) takes ~24k cycles during which gemm_dma.h's workgroup issues no MFMA -- and a wave
// cannot run ahead of its own stores, because on gfx950 loads and stores retire through one in-order vmcnt.
//
// Here the workgroup's 8 waves form two groups of 4 (one wave per SIMD each) that alternate roles every half-step:
//
//     half-step h:   group h&1       MAIN  runs the k-loop of ITS tile h      (128 x 192 outputs, wave tile 64 x 96)
//                    group (h+1)&1   EPI   converts + stores ITS tile h-1 and prefetches the first k-tile of tile h+1
//
// Both roles execute exactly NK barriers per half-step, so the hardware barrier keeps the two groups in anti-phase: the
// stores of one group drain while the other group owns the MFMA pipe.  (Two independent workgroups per CU do not stay in
// anti-phase by themselves: the shared MFMA pipe / store path make in-phase the attractor.)
// LDS: X (first k-tile of a tile, prefetched by the next MAIN group while it is still EPI, BEFORE its stores: a DMA
// queued behind stores would wait for their drain) + a ring of two stages for k-tiles 1..NK-1 = 3 x 40 KB.
// Workgroups are persistent (one per CU); tile order is XCD-aware as in gemm_dma.h.
#pragma once
#include <type_traits>
#include <utility>
#include "gemm_dma.h"

namespace skp {

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

typedef TileCfg<128, 192, 32, 2, 2> PPTile;      // one GROUP's tile; the workgroup has two groups = 512 threads

template <class P, int NK, class AS, class EP>
__global__ void __launch_bounds__(512) gemm_pp_kernel(const DmaArgs<P, AS, EP> g) {
    typedef typename P::T T;
    typedef PPTile TC;
    constexpr int NA = P::NA, NW = P::NW, BM = TC::BM, BN = TC::BN, BK = TC::BK;
    constexpr int RPI = 16;                              // tile rows per DMA instruction (BK = 32: 64-byte rows)
    constexpr int NI_A = NA * BM / RPI, NI_W = NW * BN / RPI, NI = NI_A + NI_W;
    constexpr int CNT = (NI + 3) / 4;                    // DMA instructions per wave per k-tile
    constexpr int STAGE = dma_stage_bytes<P, TC>();
    constexpr int A_BYTES = NA * BM * BK * 2;
    constexpr int FP = TC::FN / 2, UNITS = TC::FM * FP;  // epilogue store units: (a, fragment pair)
    static_assert(BK == 32 && NK >= 3, "k-tiles");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const X = smem;                                // k-tile 0 of the MAIN group's tile
    char* const R = smem + STAGE;                        // ring of two stages for k-tiles 1..NK-1

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int group = wave >> 2, gw = wave & 3, wm = gw >> 1, wn = gw & 1;
    const int lr = lane >> 2, lp = lane & 3;             // DMA lane -> (row in the 16-row piece, 16-byte chunk)
    const unsigned lds_base = (unsigned)(size_t)smem;

    // persistent tile sequence of this workgroup: virtual workgroup ids b, b + G, b + 2G, ... (same XCD), XCD-aware order
    const int ntiles = g.nM * g.nN, G = gridDim.x, b = blockIdx.x;
    const int n_seq = b < ntiles ? (ntiles - b + G - 1) / G : 0;
    auto tile_of = [&](int i, int& m_tile, int& n_tile) {
        const int v = b + i * G;
        const int q = ntiles >> 3, r = ntiles & 7, xcd = v & 7;
        const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3);
        m_tile = t / g.nN; n_tile = t - m_tile * g.nN;
    };

    // per-tile DMA sources of this wave (k-block 0), set up one half-step ahead
    const T* src[CNT];
    int dst_off[CNT];
    auto setup = [&](int i) {
        int m_tile, n_tile;
        tile_of(i, m_tile, n_tile);
        const int m0 = m_tile * BM, n0 = n_tile * BN;
#pragma unroll
        for (int c = 0; c < CNT; ++c) {
            const int q = gw + c * 4;
            src[c] = g.zrow; dst_off[c] = 0;
            if (q < NI_A) {
                const int plane = q / (BM / RPI), rb = q % (BM / RPI);
                const int r = rb * RPI + lr;
                const int chunk = lp ^ ((r >> 1) & 3);
                const T *hi, *lo;
                g.as.rows(m0 + r, chunk * 8, g.zrow, hi, lo);
                src[c] = plane == 0 ? hi : lo;
                dst_off[c] = (plane * BM + rb * RPI) * BK * 2;
            } else if (q < NI) {
                const int qw = q - NI_A;
                const int plane = qw / (BN / RPI), rb = qw % (BN / RPI);
                const int r = rb * RPI + lr;
                const int chunk = lp ^ ((r >> 1) & 3);
                const int n = n0 + r;
                src[c] = n < g.N ? g.W + blk_off(n, chunk * 8, g.ldw) + plane * g.w_plane : g.zrow;
                dst_off[c] = A_BYTES + (plane * BN + rb * RPI) * BK * 2;
            }
        }
    };
    auto issue = [&](int kt, unsigned stage_base) {
#pragma unroll
        for (int c = 0; c < CNT; ++c) {
            if (gw + c * 4 < NI) {
                const T* p = src[c];
                if (p != g.zrow) p += (long long)kt * 512;          // next 32-column block of the same row block
                glds16(p, stage_base + (unsigned)dst_off[c]);
            }
        }
    };
    auto barrier = [] { __builtin_amdgcn_s_barrier(); };

    f32x4 acc[TC::FM][TC::FN];
    int em0 = 0, en0 = 0;                                // tile whose accumulators this group holds
    bool eswap = true;

    if (group == 0 && n_seq > 0) { setup(0); issue(0, lds_base); }

    for (int h = 0; h <= n_seq; ++h) {
        if ((h & 1) == group) {
            // ---------------- MAIN: k-loop of tile h ---------------- //
            if (h < n_seq) {
                int m_tile, n_tile;
                tile_of(h, m_tile, n_tile);
                em0 = m_tile * BM; en0 = n_tile * BN;
                eswap = true;
                if constexpr (EP::kDualOrder) eswap = !g.ep.unswapped(en0);
#pragma unroll
                for (int a = 0; a < TC::FM; ++a)
#pragma unroll
                    for (int bb = 0; bb < TC::FN; ++bb) acc[a][bb] = f32x4{0.f, 0.f, 0.f, 0.f};
                for (int kt = 0; kt < NK; ++kt) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    barrier();
                    if (kt + 1 < NK) issue(kt + 1, lds_base + (unsigned)(STAGE + ((kt + 1) & 1) * STAGE));
                    const char* st = kt == 0 ? X : R + (kt & 1) * STAGE;
                    if constexpr (EP::kDualOrder) {
                        if (eswap) mma_tile<P, TC, true>(st, st + A_BYTES, acc, wm, wn, lane);
                        else       mma_tile<P, TC, false>(st, st + A_BYTES, acc, wm, wn, lane);
                    } else {
                        mma_tile<P, TC, true>(st, st + A_BYTES, acc, wm, wn, lane);
                    }
                }
            } else {
                for (int kt = 0; kt < NK; ++kt) barrier();
            }
        } else {
            // ---------------- EPI: tile h-1 out, first k-tile of tile h+1 in ---------------- //
#ifdef SKP_DEBUG_NOEPI
            const bool has_tile = false, has_next = h + 1 < n_seq;
#else
            const bool has_tile = h >= 1, has_next = h + 1 < n_seq;
#endif
            const int m0w = em0 + wm * TC::WTM, n0w = en0 + wn * TC::WTN;
            barrier();                                                  // it 0
            if (has_next) setup(h + 1);                                 // table lookups before any store is queued
            auto epi = [&](auto swap_c) {
                constexpr bool SW = decltype(swap_c)::value;
                if (has_tile) g.ep.template pp_begin<TC, SW>(acc, m0w, n0w, lane, g.M, g.N);
                static_for<1, NK>([&](auto kt_c) {
                    constexpr int kt = decltype(kt_c)::value;
                    barrier();
                    if constexpr (kt == 1) { if (has_next) issue(0, lds_base); }     // X was consumed in iteration 0
                    // the CU's memory pipeline is one in-order queue: let the MAIN group's DMA of this iteration enter it
                    // before this iteration's stores, which then drain while the MFMAs run
                    if (has_tile) {
                        static_for<0, UNITS>([&](auto u_c) {
                            constexpr int u = decltype(u_c)::value;
                            if constexpr (1 + u * (NK - 1) / UNITS == kt) {
                                for (int d = 0; d < g.pp_delay; ++d) __builtin_amdgcn_s_sleep(1);      // store pacing
                                g.ep.template pp_unit<TC, SW, u / FP, u % FP>(acc, m0w, n0w, lane, g.M, g.N);
                            }
                        });
                    }
                });
            };
            if constexpr (EP::kDualOrder) {
                if (eswap) epi(std::true_type{}); else epi(std::false_type{});
            } else {
                epi(std::true_type{});
            }
        }
    }
}

template <class P, int NK, class AS, class EP>
inline hipError_t launch_gemm_pp(DmaArgs<P, AS, EP> g, hipStream_t stream) {
    typedef PPTile TC;
    g.nM = (g.M + TC::BM - 1) / TC::BM;
    g.nN = (g.N + TC::BN - 1) / TC::BN;
    if (g.nM == 0 || g.nN == 0) return hipSuccess;
    if (g.K != NK * TC::BK || g.N % TC::BN != 0) return hipErrorInvalidValue;
    constexpr int smem = 3 * dma_stage_bytes<P, TC>();
    static_assert(smem <= 160 * 1024, "LDS per block");
    auto kern = gemm_pp_kernel<P, NK, AS, EP>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return e;
    static const int delay = getenv("SKP_PP_DELAY") ? atoi(getenv("SKP_PP_DELAY")) : 4;
    g.pp_delay = delay;
    const int tiles = g.nM * g.nN;
    const int grid = tiles < 256 ? (tiles + 7) / 8 * 8 : 256;      // one persistent workgroup per CU; multiple of 8 (XCD order)
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), smem, stream, g);
    return hipGetLastError();
}

}  // namespace skp
