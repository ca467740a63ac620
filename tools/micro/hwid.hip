// Where do the waves of a workgroup land?  Every wave records HW_REG_HW_ID / HW_REG_XCC_ID and s_memtime at its start:
//   hipcc --offload-arch=gfx950 -O2 tools/micro/hwid.hip -o tools/micro/hwid && tools/micro/hwid <threads per block> <blocks> <lds bytes>
// Output per block: XCC, SE, CU and the SIMD of each wave -- the pairing of waves on a SIMD that fused_block2.hip's skew relies on,
// and the block -> CU order that decides which two workgroups share a CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void probe(unsigned* out, int spin) {
    extern __shared__ char smem[];
    const unsigned hw = __builtin_amdgcn_s_getreg(4 | (31 << 11));
    const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (31 << 11));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6);
        out[4 * w] = hw; out[4 * w + 1] = xcc; out[4 * w + 2] = (unsigned)(t0 & 0xffffffffu); out[4 * w + 3] = (unsigned)(t0 >> 32);
    }
    // keep the block resident for a while so that later blocks have to queue behind it
    volatile float acc = threadIdx.x;
    for (int i = 0; i < spin; ++i) acc = acc * 1.0001f + 0.5f;
    if (acc == 12345.f) smem[0] = 1;
}

int main(int argc, char** argv) {
    const int threads = argc > 1 ? atoi(argv[1]) : 512, blocks = argc > 2 ? atoi(argv[2]) : 1024, lds = argc > 3 ? atoi(argv[3]) : 139776;
    const int waves = blocks * threads / 64;
    unsigned* d;
    hipMalloc(&d, waves * 16);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipLaunchKernelGGL(probe, dim3(blocks), dim3(threads), lds, 0, d, 200000);
    hipDeviceSynchronize();
    std::vector<unsigned> h(waves * 4);
    hipMemcpy(h.data(), d, waves * 16, hipMemcpyDeviceToHost);
    const int wpb = threads / 64;
    unsigned long long tmin = ~0ull;
    for (int w = 0; w < waves; ++w) { unsigned long long t = ((unsigned long long)h[4 * w + 3] << 32) | h[4 * w + 2]; if (t < tmin) tmin = t; }
    for (int b = 0; b < blocks; ++b) {
        if (!(b < 24 || (b >= 248 && b < 272) || (b >= 504 && b < 528) || b % 97 == 0)) continue;
        const unsigned hw = h[4 * b * wpb], xcc = h[4 * b * wpb + 1] & 15;
        unsigned long long t = ((unsigned long long)h[4 * b * wpb + 3] << 32) | h[4 * b * wpb + 2];
        printf("block %4d xcc %u se %u sh %u cu %2u t %8llu  simd:", b, xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15, t - tmin);
        for (int w = 0; w < wpb; ++w) printf(" %u", (h[4 * (b * wpb + w)] >> 4) & 3);
        printf("  wave_id:");
        for (int w = 0; w < wpb; ++w) printf(" %u", h[4 * (b * wpb + w)] & 15);
        printf("\n");
    }
    return 0;
}
