// Microbenchmark: HBM write bandwidth of the GEMM epilogue store patterns vs fully coalesced stores.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// each wave owns a 64-row x 96-col fp32 sub-tile of a [rows][ld] matrix, like the GEMM epilogue (FM=4, FN=6)
// MODE 0: lane (l15,g) writes float4 at row 16a+l15, col 16b+4g   (current pattern: 16 rows x 64 B per instruction)
// MODE 1: fully coalesced: the wave writes its 64x96 tile as contiguous 1 KiB pieces (what an LDS-transposed epilogue could do
//         if the tile were contiguous in memory) -- upper bound
// MODE 2: row-contiguous: per instruction 2 rows x 384 B... lane writes float4, 24 lanes per row (96 cols), i.e. rows of the tile written whole
// MODE 3: blocked 16-bit planes pattern: lane writes 8 B at block(row16, col32): 16 rows x 32 B per instr, two planes
template <int MODE>
__global__ void __launch_bounds__(512) st_kernel(float* __restrict__ out, int ld, long long rows_total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    // block tile: 128 rows x 384 cols (8 waves as 2 x 4), like D384
    const long long row0 = (long long)blockIdx.x * 128 + (wave >> 2) * 64;
    const int col0 = (wave & 3) * 96;
    if (row0 + 64 > rows_total) return;
    const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
    if (MODE == 0) {
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 6; ++b) *reinterpret_cast<float4*>(out + (row0 + a * 16 + l15) * ld + col0 + b * 16 + g * 4) = v;
    } else if (MODE == 1) {
        float* base = out + (row0 * ld) + (long long)(wave & 3) * 64 * 96;   // pretend tile-contiguous storage
#pragma unroll
        for (int i = 0; i < 24; ++i) *reinterpret_cast<float4*>(base + i * 256 + lane * 4) = v;
    } else if (MODE == 2) {
        // 24 lanes cover one row's 96 cols; 64 lanes -> 2.67 rows: use 48 lanes = 2 rows per instruction, 32 instructions
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const int r = i * 2 + (lane / 24), c = (lane % 24) * 4;
            if (lane < 48) *reinterpret_cast<float4*>(out + (row0 + r) * ld + col0 + c) = v;
        }
    } else {
        unsigned short* o16 = reinterpret_cast<unsigned short*>(out);
        const long long plane = rows_total * ld;
        const uint2 w = make_uint2(0x3f803f80u, 0x3f803f80u);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 6; ++b) {
                const long long row = row0 + a * 16 + l15; const int col = col0 + b * 16 + g * 4;
                const long long off = (((row >> 4) * (ld >> 5) + (col >> 5)) << 9) + ((row & 15) << 5) + (col & 31);
                *reinterpret_cast<uint2*>(o16 + off) = w;
                *reinterpret_cast<uint2*>(o16 + plane + off) = w;
            }
    }
}

template <int MODE>
int run(const char* name, float* d, long long rows, int ld) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int blocks = (int)(rows / 128);
    hipLaunchKernelGGL(st_kernel<MODE>, dim3(blocks), dim3(512), 0, 0, d, ld, rows);
    CK(hipEventRecord(a));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(st_kernel<MODE>, dim3(blocks), dim3(512), 0, 0, d, ld, rows);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double bytes = (double)rows * ld * 4 * 10;
    printf("%-52s %8.3f ms/launch  %6.2f TB/s\n", name, ms / 10, bytes / ms / 1e9);
    return 0;
}

int main() {
    const long long rows = 131072 * 4; const int ld = 384;        // 805 MB per launch
    float* d; CK(hipMalloc(&d, rows * ld * 4 + (1 << 20)));
    run<1>("coalesced 1 KiB per instruction (upper bound)", d, rows, ld);
    run<0>("epilogue pattern: 16 rows x 64 B (fp32 row-major)", d, rows, ld);
    run<2>("row-contiguous: 2 rows x 384 B per instruction", d, rows, ld);
    run<3>("blocked 16-bit planes: 16 rows x 32 B, 2 planes", d, rows, ld);
    return 0;
}
