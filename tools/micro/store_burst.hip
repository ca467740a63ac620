// Microbenchmark: what bounds the GEMM epilogue's store burst?  Every workgroup (8 waves) writes one "tile" of
// 192 KiB as 16-byte-per-lane stores (one contiguous 1 KiB per wave instruction), like the fc1 epilogue after perm8.
// Swept over the number of workgroups in flight (per-CU limit vs chip limit) and tiles per workgroup.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int W16>   // W16 = 1: 16-byte stores, 0: 8-byte stores (two per 16 B)
__global__ void __launch_bounds__(512) burst(char* __restrict__ out, int tiles_per_wg) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint4 v = make_uint4(lane, wave, blockIdx.x, 7u);
    for (int t = 0; t < tiles_per_wg; ++t) {
        char* base = out + ((long long)blockIdx.x * tiles_per_wg + t) * (192 * 1024) + wave * (24 * 1024);
#pragma unroll
        for (int i = 0; i < 24; ++i) {
            if (W16) *reinterpret_cast<uint4*>(base + i * 1024 + lane * 16) = v;
            else {
                *reinterpret_cast<uint2*>(base + i * 1024 + (lane & 15) * 64 + (lane >> 4) * 8) = make_uint2(v.x, v.y);
                *reinterpret_cast<uint2*>(base + i * 1024 + (lane & 15) * 64 + (lane >> 4) * 8 + 32) = make_uint2(v.z, v.w);
            }
        }
    }
}

template <int W16>
int run(char* d, int wgs, int tiles) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(burst<W16>, dim3(wgs), dim3(512), 0, 0, d, tiles);
    CK(hipEventRecord(a));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(burst<W16>, dim3(wgs), dim3(512), 0, 0, d, tiles);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double bytes = (double)wgs * tiles * 192 * 1024 * 5;
    const int cus = wgs < 256 ? wgs : 256;
    printf("%s  wgs %5d tiles/wg %3d: %8.3f ms/launch  %6.2f TB/s  %6.1f B/clk/CU(2.4GHz, %d CUs)\n", W16 ? "16B" : " 8B", wgs, tiles, ms / 5,
           bytes / ms / 1e9, bytes / 5 / (ms * 1e-3) / cus / 2.4e9, cus);
    return 0;
}

int main() {
    char* d; CK(hipMalloc(&d, (size_t)8192 * 192 * 1024 + (1 << 20)));
    for (int wgs : {32, 64, 128, 256}) { run<1>(d, wgs, 32); }
    run<1>(d, 8192, 1); run<0>(d, 8192, 1); run<0>(d, 256, 32); run<0>(d, 64, 32);
    return 0;
}
