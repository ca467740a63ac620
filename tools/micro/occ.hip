#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(256) k256(float* p) { extern __shared__ char s[]; if (p) p[threadIdx.x] = s[threadIdx.x]; }
__global__ void __launch_bounds__(512) k512(float* p) { extern __shared__ char s[]; if (p) p[threadIdx.x] = s[threadIdx.x]; }
int main() {
    int v; hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, 0); printf("MaxSharedMemoryPerMultiprocessor %d\n", v);
    hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, 0); printf("MaxSharedMemoryPerBlock %d\n", v);
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0); printf("sharedMemPerBlock %zu sharedMemPerMultiprocessor %zu maxSharedMemoryPerMultiProcessor %zu\n", pr.sharedMemPerBlock, pr.sharedMemPerMultiprocessor, pr.maxSharedMemoryPerMultiProcessor);
    for (int kb : {32, 48, 64, 72, 76, 78, 79, 80, 81, 96, 112, 128, 160}) {
        int n256 = -1, n512 = -1;
        hipFuncSetAttribute((const void*)k256, hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024);
        hipFuncSetAttribute((const void*)k512, hipFuncAttributeMaxDynamicSharedMemorySize, kb * 1024);
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&n256, k256, 256, kb * 1024);
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&n512, k512, 512, kb * 1024);
        printf("LDS %3d KB: blocks/CU 256thr=%d 512thr=%d\n", kb, n256, n512);
    }
    return 0;
}
