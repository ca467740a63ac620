// How fast does ONE wave per SIMD issue v_mfma_f32_32x32x16_f16?  NCHAIN independent accumulator chains, operands random or constant.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/mfma32_rate.hip -o tools/micro/mfma32_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NCHAIN, int WPE, bool RANDOM>
__global__ void __launch_bounds__(256 * WPE) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) k(float* out, long long* clk, int iters) {
    f32x16 acc[NCHAIN];
    f16x8 a[4], b[4];
    unsigned r = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int q = 0; q < 4; ++q)
        for (int i = 0; i < 8; ++i) {
            r = r * 1664525u + 1013904223u;
            a[q][i] = RANDOM ? (_Float16)(((int)(r >> 20) - 2048) / 4096.0f) : (_Float16)0.5f;
            r = r * 1664525u + 1013904223u;
            b[q][i] = RANDOM ? (_Float16)(((int)(r >> 20) - 2048) / 65536.0f) : (_Float16)0.25f;
        }
    for (int c = 0; c < NCHAIN; ++c)
        for (int v = 0; v < 16; ++v) acc[c][v] = 0.f;
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int c = 0; c < NCHAIN; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[q], b[(q + c) & 3], acc[c], 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int c = 0; c < NCHAIN; ++c) s += acc[c][0] + acc[c][15];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int NCHAIN, int WPE, bool RANDOM>
void run(int grid, const char* name) {
    float* out; long long* clk;
    hipMalloc(&out, (size_t)grid * 256 * WPE * 4); hipMalloc(&clk, grid * 8);
    const int iters = 2000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NCHAIN, WPE, RANDOM>), dim3(grid), dim3(256 * WPE), 0, 0, out, clk, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NCHAIN, WPE, RANDOM>), dim3(grid), dim3(256 * WPE), 0, 0, out, clk, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c0; hipMemcpy(&c0, clk, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * 4 * NCHAIN;
    printf("%-36s grid %4d: %6.1f counter ticks / MFMA / wave; wall %.3f ms -> %.1f ns per MFMA per SIMD, %.0f TFLOP/s\n", name, grid, c0 / n, ms,
           ms * 1e6 / (n * WPE), grid * 4.0 * WPE * n * 32768 / (ms * 1e-3) / 1e12);
    hipFree(out); hipFree(clk);
}


template <int NCHAIN, int WPE, bool RANDOM>
__global__ void __launch_bounds__(256 * WPE) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) k16(float* out, long long* clk, int iters) {
    f32x4 acc[NCHAIN];
    f16x8 a[4], b[4];
    unsigned r = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int q = 0; q < 4; ++q)
        for (int i = 0; i < 8; ++i) {
            r = r * 1664525u + 1013904223u;
            a[q][i] = RANDOM ? (_Float16)(((int)(r >> 20) - 2048) / 4096.0f) : (_Float16)0.5f;
            r = r * 1664525u + 1013904223u;
            b[q][i] = RANDOM ? (_Float16)(((int)(r >> 20) - 2048) / 65536.0f) : (_Float16)0.25f;
        }
    for (int c = 0; c < NCHAIN; ++c) acc[c] = f32x4{0, 0, 0, 0};
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int c = 0; c < NCHAIN; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[q], b[(q + c) & 3], acc[c], 0, 0, 0);
    }
    const long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int c = 0; c < NCHAIN; ++c) s += acc[c][0] + acc[c][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}
template <int NCHAIN, int WPE, bool RANDOM>
void run16(int grid, const char* name) {
    float* out; long long* clk;
    hipMalloc(&out, (size_t)grid * 256 * WPE * 4); hipMalloc(&clk, grid * 8);
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k16<NCHAIN, WPE, RANDOM>), dim3(grid), dim3(256 * WPE), 0, 0, out, clk, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k16<NCHAIN, WPE, RANDOM>), dim3(grid), dim3(256 * WPE), 0, 0, out, clk, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c0; hipMemcpy(&c0, clk, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * 4 * NCHAIN;
    printf("16x16x32 %-28s grid %4d: %6.1f ticks / MFMA / wave; wall %.3f ms, %.0f TFLOP/s\n", name, grid, c0 / n, ms,
           grid * 4.0 * WPE * n * 16384 / (ms * 1e-3) / 1e12);
    hipFree(out); hipFree(clk);
}

int main() {
    run16<4, 2, false>(256, "4 chains, 2 waves, const");
    run16<4, 2, true>(256, "4 chains, 2 waves, random");
    run16<8, 1, true>(256, "8 chains, 1 wave, random");
    run16<2, 2, true>(256, "2 chains, 2 waves, random");
    for (int grid : {256}) {
        run<1, 1, false>(grid, "1 chain, 1 wave/SIMD, const");
        run<2, 1, false>(grid, "2 chains, 1 wave/SIMD, const");
        run<4, 1, false>(grid, "4 chains, 1 wave/SIMD, const");
        run<12, 1, false>(grid, "12 chains, 1 wave/SIMD, const");
        run<1, 1, true>(grid, "1 chain, 1 wave/SIMD, random");
        run<2, 1, true>(grid, "2 chains, 1 wave/SIMD, random");
        run<4, 1, true>(grid, "4 chains, 1 wave/SIMD, random");
        run<12, 1, true>(grid, "12 chains, 1 wave/SIMD, random");
        run<4, 2, true>(grid, "4 chains, 2 waves/SIMD, random");
    }
    return 0;
}
