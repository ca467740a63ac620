// Microbenchmark: can one CU drain stores from 4 waves while 4 other waves (one per SIMD) run an LDS-DMA + MFMA loop?
// MODE bit 0: waves 0-3 run the "main loop" (10 LDS-DMA loads of 1 KiB + vmcnt(0) + 72 MFMAs per iteration)
// MODE bit 1: waves 4-7 run the "epilogue" (stores of 1 KiB per wave instruction, STORES_PER_ITER per main iteration)
// No barriers: any slowdown of one role by the other is the hardware's (shared memory pipeline), not the schedule's.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_dst);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}

__global__ void __launch_bounds__(512) k(const char* __restrict__ in, char* __restrict__ out, float* sink, int mode, int iters, int stores_per_iter) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave < 4) {
        if (!(mode & 1)) return;
        f32x4 acc[24];
        for (int i = 0; i < 24; ++i) acc[i] = f32x4{0, 0, 0, 0};
        f16x8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {1, 1, 1, 1, 1, 1, 1, 1};
        const unsigned lds = (unsigned)(size_t)smem + wave * 20480;
        const char* src = in + ((size_t)blockIdx.x * 4 + wave) * 20480 * 4 + lane * 16;
        for (int it = 0; it < iters; ++it) {
            for (int c = 0; c < 10; ++c) glds16(src + ((it & 3) * 10 + c) * 1024, lds + ((it & 1) * 10 + c) * 1024);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int i = 0; i < 24; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
        }
        float s = 0;
        for (int i = 0; i < 24; ++i) s += acc[i][0];
        if (s == 123.f) sink[0] = s;
    } else {
        if (!(mode & 2)) return;
        const uint4 v = make_uint4(lane, wave, blockIdx.x, 7u);
        char* base = out + ((size_t)blockIdx.x * 4 + (wave - 4)) * ((size_t)iters * stores_per_iter * 1024) + lane * 16;
        for (int it = 0; it < iters; ++it)
            for (int s = 0; s < stores_per_iter; ++s) *reinterpret_cast<uint4*>(base + ((size_t)it * stores_per_iter + s) * 1024) = v;
    }
}

int main() {
    const int iters = 2000;
    char *in, *out; float* sink;
    CK(hipMalloc(&in, (size_t)256 * 4 * 20480 * 4 + 4096));
    CK(hipMalloc(&out, (size_t)256 * 4 * iters * 8 * 1024 + 4096));
    CK(hipMalloc(&sink, 64));
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int spi : {2, 3, 4, 6}) {
        for (int mode : {1, 2, 3}) {
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 120 * 1024, 0, in, out, sink, mode, iters, spi);
            CK(hipEventRecord(a));
            hipLaunchKernelGGL(k, dim3(256), dim3(512), 120 * 1024, 0, in, out, sink, mode, iters, spi);
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b));
            printf("stores/iter/wave %d  mode %d (%s): %7.3f ms   (%.0f cycles per iteration at 2.4 GHz; stores %.1f KB/iter/CU)\n", spi, mode,
                   mode == 1 ? "main only" : mode == 2 ? "stores only" : "both", ms, ms * 1e-3 * 2.4e9 / iters, spi * 4.0);
        }
    }
    return 0;
}
