// Delivery helpers (include/skyrim_io.h): the byte swap netCDF-3 asks of every float, done where the state already is.
// HBM-bound, one pass: 16-byte loads and stores while the pointers allow it, a scalar tail.
#include <hip/hip_runtime.h>
#include <cstdint>
#include "../../include/skyrim_io.h"

namespace {

__device__ __forceinline__ uint32_t swap32(uint32_t v) { return __builtin_bswap32(v); }

__global__ void __launch_bounds__(256) bswap32_vec_kernel(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n_vec) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_vec; i += stride) {
        uint4 v = src[i];
        v.x = swap32(v.x); v.y = swap32(v.y); v.z = swap32(v.z); v.w = swap32(v.w);
        dst[i] = v;
    }
}

__global__ void bswap32_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = swap32(src[i]);
}

}  // namespace

extern "C" int skio_abi_version(void) { return SKIO_ABI_VERSION; }

extern "C" int skio_bswap32(const void* src, void* dst, size_t n_words, void* stream) {
    if (n_words == 0) return 0;
    if (!src || !dst || (((uintptr_t)src | (uintptr_t)dst) & 3)) return SKIO_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    const uint32_t* sp = (const uint32_t*)src;
    uint32_t* dp = (uint32_t*)dst;
    size_t done = 0;
    if ((((uintptr_t)src | (uintptr_t)dst) & 15) == 0 && n_words >= 4) {
        const size_t n_vec = n_words / 4;
        // 256 CUs x 8 workgroups of 256 lanes in flight; every lane walks the array with the grid's stride (coalesced 16-byte accesses)
        const size_t want = (n_vec + 255) / 256;
        const unsigned blocks = (unsigned)(want < 2048 ? want : 2048);
        hipLaunchKernelGGL(bswap32_vec_kernel, dim3(blocks), dim3(256), 0, s, (const uint4*)src, (uint4*)dst, n_vec);
        done = n_vec * 4;
    }
    if (done < n_words) {
        const size_t rest = n_words - done;
        hipLaunchKernelGGL(bswap32_kernel, dim3((unsigned)((rest + 255) / 256)), dim3(256), 0, s, sp + done, dp + done, rest);
    }
    return hipGetLastError() == hipSuccess ? 0 : SKIO_E_HIP;
}
