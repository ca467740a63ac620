// How fast does ONE wave issue v_mfma_f32_16x16x32_f16?  Clocks per MFMA for NCHAIN independent accumulator chains, 1 or 2 waves per SIMD,
// on one CU or on all of them (the sustained clock differs); wall clock via events gives the chip-wide rate.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/mfma_rate.hip -o tools/micro/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NCHAIN, int WPE>
__global__ void __launch_bounds__(256 * WPE) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) k(float* out, long long* clk, int iters) {
    f32x4 acc[NCHAIN];
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f); }
    for (int c = 0; c < NCHAIN; ++c) acc[c] = f32x4{0, 0, 0, 0};
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < NCHAIN; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[c], 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int c = 0; c < NCHAIN; ++c) s += acc[c][0] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int NCHAIN, int WPE>
void run(int grid, const char* name) {
    float* out; long long* clk;
    hipMalloc(&out, (size_t)grid * 256 * WPE * 4); hipMalloc(&clk, grid * 8);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NCHAIN, WPE>), dim3(grid), dim3(256 * WPE), 0, 0, out, clk, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NCHAIN, WPE>), dim3(grid), dim3(256 * WPE), 0, 0, out, clk, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c0; hipMemcpy(&c0, clk, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * 4 * NCHAIN;
    printf("%-28s grid %4d: %6.1f counter ticks / MFMA / wave; wall %.3f ms -> %.1f ns per MFMA per SIMD-slot, %.0f TFLOP/s\n", name, grid, c0 / n, ms,
           ms * 1e6 / (n * WPE), grid * 4.0 * WPE * n * 16384 / (ms * 1e-3) / 1e12);
    hipFree(out); hipFree(clk);
}

int main() {
    for (int grid : {1, 256}) {
        run<1, 1>(grid, "1 chain, 1 wave/SIMD");
        run<2, 1>(grid, "2 chains, 1 wave/SIMD");
        run<4, 1>(grid, "4 chains, 1 wave/SIMD");
        run<8, 1>(grid, "8 chains, 1 wave/SIMD");
        run<16, 1>(grid, "16 chains, 1 wave/SIMD");
        run<1, 2>(grid, "1 chain, 2 waves/SIMD");
        run<2, 2>(grid, "2 chains, 2 waves/SIMD");
        run<4, 2>(grid, "4 chains, 2 waves/SIMD");
        run<8, 2>(grid, "8 chains, 2 waves/SIMD");
    }
    return 0;
}
