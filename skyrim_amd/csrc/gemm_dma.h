// MFMA GEMM whose operands are BOTH pre-split 16-bit planes in HBM and reach LDS by LDS-DMA
// (global_load_lds_dwordx4): no VGPR staging, no conversion VALU and no ds_write in the main loop.
//
//   out = epilogue( A[M,K] * W[N,K]^T ),   A rows = rows of hi/lo planes (optionally gathered through a table)
//
// Pipeline: two LDS stages; iteration kt waits for its own DMA (vmcnt(0)) + one barrier, issues the
// DMA of tile kt+1 into the other stage and runs the MFMAs of tile kt while that DMA is in flight.
// The LDS image of a tile is lane-linear per DMA instruction (64 lanes x 16 B = 1 KiB = RPI tile rows), so the
// bank-conflict swizzle of common.h::lds_off is applied on the SOURCE address: lane L of the instruction
// covering tile rows [r0, r0+RPI) writes physical chunk p = L % CPR of row r = r0 + L / CPR and therefore
// fetches logical chunk p ^ f(r).  Rows that do not exist (window padding, M tail) read a zero row.
// Workgroups are renumbered so that the n-tiles of one m-tile run back to back on ONE XCD (A tile L2 reuse).
#pragma once
#include "gemm.h"

namespace skp {

typedef __attribute__((address_space(3))) void lds_void_t;
typedef const __attribute__((address_space(1))) void glb_void_t;

// ---- A-operand sources: per-row pointers to the hi / lo plane ------------------ //
template <class T>
struct APlanes {
    const T* x;            // hi plane [rows][ld]
    long long plane;       // lo plane offset (elements)
    int ld;
    const int* idx;        // optional gather table (window partition); -1 = padding row
    int M;
    __device__ __forceinline__ void rows(int m, int k0, const T* zrow, const T*& hi, const T*& lo) const {
        int s = -1;
        if (m < M) s = idx ? idx[m] : m;
        if (s < 0) { hi = zrow; lo = zrow; return; }
        hi = x + blk_off(s, k0, ld);
        lo = hi + plane;
    }
    static constexpr int kSplitK = 0;
    __device__ __forceinline__ long long second_half_delta() const { return 0; }
};

// concat(a, b) along K (PatchRecovery's concat(skip, x)): columns [0,C1) from a, [C1,2*C1) from b
template <class T>
struct AConcatPlanes {
    const T* a; const T* b;
    long long plane;       // same plane stride for both
    int C1, M, row_off;
    __device__ __forceinline__ void rows(int m, int k0, const T* zrow, const T*& hi, const T*& lo) const {
        if (m >= M) { hi = zrow; lo = zrow; return; }
        hi = a + blk_off(m + row_off, k0, C1);
        lo = hi + plane;
    }
    static constexpr int kSplitK = 1;
    // pointer adjustment once k crosses C1: from a's column block C1/32 to b's column block 0
    __device__ __forceinline__ long long second_half_delta() const { return (long long)(b - a) - (long long)(C1 >> 5) * 512; }
};

// One LDS-DMA instruction: 64 lanes x 16 B from per-lane global addresses to LDS [lds_dst, lds_dst + 1 KiB).
// Issued through inline asm on purpose: hipcc's waitcnt pass cannot prove that the DMA's destination stage and the
// stage the MFMA loop is reading are different LDS ranges, and would put s_waitcnt vmcnt(0) in front of the
// first ds_read of every k-tile (serialising load and compute).  The kernel counts these loads itself
// (one explicit vmcnt(0) + barrier per k-tile); there is no other VMEM traffic inside the main loop.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_dst);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}

// The same with a wave-uniform base address in SGPRs and ONE per-lane byte offset (lane * 16 for a contiguous KiB): no 64-bit
// per-lane address arithmetic, no address VGPR pair per request.
__device__ __forceinline__ void glds16_s(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_dst);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(dst) : "memory");
}

template <class P, class AS, class EP>
struct DmaArgs {
    AS as;
    EP ep;
    const typename P::T* W;
    long long w_plane;
    int ldw;
    const typename P::T* zrow;   // >= max K zeros
    int M, N, K;                 // K % BK == 0
    int nM, nN;                  // tile counts
};

template <class P, class TC>
constexpr int dma_stage_bytes() { return (P::NA * TC::BM + P::NW * TC::BN) * TC::BK * 2; }

template <class P, class TC, bool SWAP>
__device__ __forceinline__ void mma_tile(const char* As, const char* Ws, f32x4 (&acc)[TC::FM][TC::FN], int wm, int wn, int lane) {
    typedef typename P::T T;
    constexpr int NA = P::NA, NW = P::NW, BK = TC::BK, FM = TC::FM, FN = TC::FN;
    constexpr int A_PLANE = TC::BM * BK * 2, W_PLANE = TC::BN * BK * 2;
    const int fr_row = lane & 15, fr_grp = lane >> 4;
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
}

template <class P, class TC, class AS, class EP, bool SWAP>
__device__ __forceinline__ void gemm_dma_body(const DmaArgs<P, AS, EP>& g, char* smem, int m_tile, int n_tile) {
    typedef typename P::T T;
    constexpr int NA = P::NA, NW = P::NW, BM = TC::BM, BN = TC::BN, BK = TC::BK, CPR = TC::CPR;
    constexpr int RPI = 64 / CPR;                       // tile rows per DMA instruction
    constexpr int NWAVES = TC::THREADS / 64;
    constexpr int NI_A = NA * BM / RPI, NI_W = NW * BN / RPI, NI = NI_A + NI_W;
    constexpr int CNT = (NI + NWAVES - 1) / NWAVES;     // DMA instructions per wave per k-tile
    constexpr int STAGE = dma_stage_bytes<P, TC>();
    constexpr int A_BYTES = NA * BM * BK * 2;
    static_assert(BM % RPI == 0 && BN % RPI == 0, "tile rows per DMA instruction");

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / TC::WN, wn = wave % TC::WN;
    const int m0 = m_tile * BM, n0 = n_tile * BN;

    // this lane's source pointers, one per DMA instruction of its wave (swizzled chunk folded in)
    const T* src[CNT];
    int dst_off[CNT];
    bool is_a[CNT];
    const int lr = lane / CPR, lp = lane % CPR;
#pragma unroll
    for (int i = 0; i < CNT; ++i) {
        const int q = wave + i * NWAVES;
        src[i] = g.zrow; dst_off[i] = 0; is_a[i] = false;
        if (q < NI_A) {
            const int plane = q / (BM / RPI), rb = q % (BM / RPI);
            const int r = rb * RPI + lr;
            const int chunk = (BK == 64) ? (lp ^ (r & 7)) : (lp ^ ((r >> 1) & 3));
            const T *hi, *lo;
            // un-swapped tiles (V of the QKV linear): tile rows in perm8 order, so that a lane's accumulators of a fragment
            // pair along M are 8 consecutive tokens (16-byte stores into V^T, epilogues.h EpQKV)
            g.as.rows(m0 + ((EP::kDualOrder && !SWAP) ? perm8_col(r) : r), chunk * 8, g.zrow, hi, lo);
            src[i] = plane == 0 ? hi : lo;
            dst_off[i] = (plane * BM + rb * RPI) * BK * 2;
            is_a[i] = true;
        } else if (q < NI) {
            const int qw = q - NI_A;
            const int plane = qw / (BN / RPI), rb = qw % (BN / RPI);
            const int r = rb * RPI + lr;
            const int chunk = (BK == 64) ? (lp ^ (r & 7)) : (lp ^ ((r >> 1) & 3));
            const int n = n0 + r;
            src[i] = n < g.N ? g.W + blk_off(n, chunk * 8, g.ldw) + plane * g.w_plane : g.zrow;
            dst_off[i] = A_BYTES + (plane * BN + rb * RPI) * BK * 2;
        }
    }
    const long long a_delta = g.as.second_half_delta();
    const int k_split = AS::kSplitK ? g.K / 2 : 0x7fffffff;

    f32x4 acc[TC::FM][TC::FN];
#pragma unroll
    for (int a = 0; a < TC::FM; ++a)
#pragma unroll
        for (int b = 0; b < TC::FN; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const unsigned lds_base = (unsigned)(size_t)smem;     // LDS byte address of the dynamic region (low 32 bits of the flat pointer)
    auto issue = [&](int kt, int stage) {
        const int k = kt * BK;
#pragma unroll
        for (int i = 0; i < CNT; ++i) {
            if (wave + i * NWAVES < NI) {
                const T* p = src[i];
                const bool zero = (p == g.zrow);
                if (!zero) {
                    p += (long long)(k >> 5) * 512;      // next column block(s) of the same row block
                    if (AS::kSplitK && is_a[i] && k >= k_split) p += a_delta;
                }
                glds16(p, lds_base + (unsigned)(stage * STAGE + dst_off[i]));
            }
        }
    };

    const int nk = g.K / BK;
    issue(0, 0);
    g.ep.template init<TC>(smem + 2 * STAGE + kEpiReduceBytes, tid, n0);     // LDS tables; ordered by the loop's first barrier
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nk) issue(kt + 1, (kt + 1) & 1);
        const char* st = smem + (kt & 1) * STAGE;
        mma_tile<P, TC, SWAP>(st, st + A_BYTES, acc, wm, wn, lane);
    }
    __syncthreads();
    g.ep.template run<TC, SWAP>(acc, m0 + wm * TC::WTM, n0 + wn * TC::WTN, lane, wm, wn, smem + 2 * STAGE, g.M, g.N, n_tile);
}

template <class P, class TC, class AS, class EP>
__global__ void __launch_bounds__(TC::THREADS) gemm_dma_kernel(const DmaArgs<P, AS, EP> g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // XCD-aware renumbering: hardware places workgroup b on XCD b % 8; give each XCD a contiguous run of tiles
    // (n fastest) so the n-tiles sharing an A tile hit the same L2.  Bijective for any grid size.
    const int nwg = gridDim.x, b = blockIdx.x;
    const int q = nwg >> 3, r = nwg & 7, xcd = b & 7;
    const int t = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    const int m_tile = t / g.nN, n_tile = t - m_tile * g.nN;
    if constexpr (EP::kDualOrder) {
        if (g.ep.unswapped(n_tile * TC::BN)) {
            gemm_dma_body<P, TC, AS, EP, false>(g, smem, m_tile, n_tile);
            return;
        }
    }
    gemm_dma_body<P, TC, AS, EP, true>(g, smem, m_tile, n_tile);
}

template <class P, class TC, class AS, class EP>
inline hipError_t launch_gemm_dma(DmaArgs<P, AS, EP> g, hipStream_t stream) {
    g.nM = (g.M + TC::BM - 1) / TC::BM;
    g.nN = (g.N + TC::BN - 1) / TC::BN;
    if (g.nM == 0 || g.nN == 0) return hipSuccess;
    if (g.K % TC::BK != 0) return hipErrorInvalidValue;
    constexpr int smem = 2 * dma_stage_bytes<P, TC>() + kEpiScratch;
    static_assert(smem <= 160 * 1024, "LDS per block");
    auto kern = gemm_dma_kernel<P, TC, AS, EP>;
    if (smem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)(g.nM * g.nN)), dim3(TC::THREADS), smem, stream, g);
    return hipGetLastError();
}

}  // namespace skp
