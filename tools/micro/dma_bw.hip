// Microbenchmark: L2-resident global -> LDS (DMA) / -> VGPR bandwidth per CU for the GEMM staging patterns.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_dst);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}

// MODE 0: DMA, 16 rows x 64 B per instruction (row stride = ld bytes)   MODE 1: DMA, contiguous 1 KiB per instruction
// MODE 2: plain global_load_dwordx4 to registers, contiguous            MODE 3: as 0 but row stride 128 B (full lines over 2 instr)
template <int MODE>
__global__ void __launch_bounds__(512) bw_kernel(const char* __restrict__ src, size_t span, int ld, int iters, float* sink) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds_base = (unsigned)(size_t)smem;
    // each block walks its own window of `span` bytes repeatedly (L2 resident)
    const char* base = src + (size_t)blockIdx.x % 64 * span;
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        const size_t tile = (size_t)(it % 8) * 65536 % span;
#pragma unroll
        for (int i = 0; i < 8; ++i) {          // 8 instr per wave per iteration -> 64 KiB per block per iteration
            const int q = wave * 8 + i;         // 0..63
            const char* p;
            if (MODE == 0)      p = base + tile % (span / 2) + (size_t)(q * 16 + (lane >> 2)) * ld + (lane & 3) * 16;
            else if (MODE == 3) p = base + tile + (size_t)(q * 16 + (lane >> 2)) * 128 + (lane & 3) * 16 + ((it & 1) * 64);
            else                p = base + tile + (size_t)q * 1024 + lane * 16;
            if (MODE == 2) { const uint4 v = *reinterpret_cast<const uint4*>(p); acc += __uint_as_float(v.x ^ v.y ^ v.z ^ v.w); }
            else glds16(p, lds_base + (it & 1) * 65536 + q * 1024);
        }
        if (MODE != 2) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        __syncthreads();
    }
    if (MODE != 2) acc = *reinterpret_cast<float*>(smem + threadIdx.x * 4);
    if (acc == 123.456f) sink[0] = acc;
}

template <int MODE>
int run(const char* name, const char* d, size_t span, int ld, float* sink) {
    const int blocks = 256, iters = 2000;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(bw_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
    hipLaunchKernelGGL(bw_kernel<MODE>, dim3(blocks), dim3(512), 131072, 0, d, span, ld, 50, sink);
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(bw_kernel<MODE>, dim3(blocks), dim3(512), 131072, 0, d, span, ld, iters, sink);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double bytes = (double)blocks * iters * 65536;
    printf("%-44s %8.3f ms  %7.2f TB/s  %6.1f GB/s/CU  (%.1f B/clk/CU @2.4GHz)\n", name, ms, bytes / ms / 1e9, bytes / ms / 1e6 / 256, bytes / ms / 1e6 / 256 / 2.4);
    return 0;
}

int main() {
    const size_t span = 1 << 20;     // 1 MiB window per block group; 64 windows = 64 MiB total (fits L2+MALL; per-XCD L2 4 MiB)
    char* d; float* sink;
    CK(hipMalloc(&d, 64 * span + (1 << 22))); CK(hipMemset(d, 1, 64 * span + (1 << 22))); CK(hipMalloc(&sink, 4));
    run<1>("DMA contiguous 1KiB/instr", d, span, 0, sink);
    run<0>("DMA 16 rows x 64B, row stride 768B", d, span, 768, sink);
    run<0>("DMA 16 rows x 64B, row stride 3072B", d, span, 3072, sink);
    run<3>("DMA 16 rows x 64B of 128B rows (alt halves)", d, span, 0, sink);
    run<2>("VGPR contiguous dwordx4", d, span, 0, sink);
    // small footprint: 64 KiB window (L1/L2 hot)
    run<1>("DMA contiguous, 128KiB window", d, 131072, 0, sink);
    run<2>("VGPR contiguous, 128KiB window", d, 131072, 0, sink);
    return 0;
}
