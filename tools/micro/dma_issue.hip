// Microbenchmark: what does ONE LDS-DMA request cost the wave that issues it, between MFMAs, when the wave has its SIMD to itself
// (one 256-thread workgroup per CU = one wave per SIMD, the shape of csrc/graphcast_fused.hip)?
//   per iteration: 8 independent v_mfma_f32_16x16x32_f16 (128 clocks of matrix pipe) + P requests of 1 KiB, P = 0 / 1 / 2
// Request forms:
//   0  none
//   1  glds16 of csrc/gemm_dma.h: save m0, set m0, s_nop, global_load_lds_dwordx4 v[addr], off, restore m0
//   2  m0 set once per iteration (no save / restore), global_load_lds_dwordx4 v[addr], off
//   3  m0 set once, global_load_lds_dwordx4 v_off, s[base:base+1]   (SGPR base + 32-bit lane offset)
//   4  m0 set once, buffer_load_dwordx4 v_off, s[rsrc], 0 offen lds
// hipcc --offload-arch=gfx950 -O3 tools/micro/dma_issue.hip -o tools/micro/dma_issue && tools/micro/dma_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int FORM>
__device__ __forceinline__ void request(const char* src, unsigned voff, const char* sbase, i32x4 rsrc, unsigned dst) {
    if constexpr (FORM == 1) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    } else if constexpr (FORM == 2) {
        asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(dst) : "memory");
    } else if constexpr (FORM == 3) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(voff), "s"(sbase), "s"(dst) : "memory");
    } else if constexpr (FORM == 4) {
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" : : "v"(voff), "s"(rsrc), "s"(dst) : "memory");
    }
}

template <int FORM, int P>
__global__ void __launch_bounds__(256) k(const char* __restrict__ in, float* sink, long long* clocks, int iters) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0, 0, 0, 0};
    f16x8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {1, 1, 1, 1, 1, 1, 1, 1};
    const unsigned lds = (unsigned)(size_t)smem + wave * 32768;
    const char* sbase = in + (size_t)(blockIdx.x & 63) * (1 << 20);           // 64 MiB window: L2 / MALL resident
    const char* src = sbase + wave * 65536 + lane * 16;
    const unsigned voff = wave * 65536 + lane * 16;
    i32x4 rsrc;
    rsrc[0] = (int)(size_t)sbase; rsrc[1] = (int)((size_t)sbase >> 32); rsrc[2] = 1 << 20; rsrc[3] = 0x00020000;
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0" : "=s"(keep));
    const long long t0 = (long long)__builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
        if (P >= 1) request<FORM>(src + (it & 31) * 1024, voff + (it & 31) * 1024, sbase, rsrc, lds + (it & 15) * 1024);
#pragma unroll
        for (int i = 4; i < 8; ++i) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
        if (P >= 2) request<FORM>(src + 32768 + (it & 31) * 1024, voff + 32768 + (it & 31) * 1024, sbase, rsrc, lds + 16384 + (it & 15) * 1024);
        if ((it & 15) == 15) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = (long long)__builtin_amdgcn_s_memtime();
    asm volatile("s_mov_b32 m0, %0" : : "s"(keep));
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0];
    if (s == 123.f) sink[0] = s + *reinterpret_cast<float*>(smem + threadIdx.x * 4);
    if (threadIdx.x == 0) clocks[blockIdx.x] = t1 - t0;
}

template <int FORM, int P>
int run(const char* name, const char* in, float* sink, long long* clocks) {
    const int iters = 4096;
    hipLaunchKernelGGL((k<FORM, P>), dim3(256), dim3(256), 131072, 0, in, sink, clocks, 64);
    hipLaunchKernelGGL((k<FORM, P>), dim3(256), dim3(256), 131072, 0, in, sink, clocks, iters);
    CK(hipDeviceSynchronize());
    std::vector<long long> h(256);
    CK(hipMemcpy(h.data(), clocks, 256 * 8, hipMemcpyDeviceToHost));
    double m = 0;
    for (auto v : h) m += (double)v;
    printf("%-70s %8.1f clocks per iteration (8 MFMAs + %d requests)\n", name, m / 256 / iters, P);
    return 0;
}

int main() {
    char* in; float* sink; long long* clocks;
    CK(hipMalloc(&in, (size_t)64 << 20)); CK(hipMemset(in, 0, (size_t)64 << 20));
    CK(hipMalloc(&sink, 64)); CK(hipMalloc(&clocks, 256 * 8));
#define RUN(F, P, NAME) CK(hipFuncSetAttribute((const void*)k<F, P>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072)); if (run<F, P>(NAME, in, sink, clocks)) return 1;
    RUN(0, 0, "no request")
    RUN(1, 1, "glds16 (save / set / restore m0, 64-bit lane addresses)")
    RUN(1, 2, "glds16 x2")
    RUN(2, 1, "m0 set, global_load_lds_dwordx4 v[addr], off")
    RUN(2, 2, "... x2")
    RUN(3, 1, "m0 set, global_load_lds_dwordx4 v_off, s[base]")
    RUN(3, 2, "... x2")
    RUN(4, 1, "m0 set, buffer_load_dwordx4 v_off, s[rsrc], 0 offen lds")
    RUN(4, 2, "... x2")
    return 0;
}
