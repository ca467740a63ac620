// Timing harness for the row-tile block kernels on synthetic buffers (no Python): one launch shape per variant index, ms per launch.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I skyrim_amd/csrc tools/micro/wide_probe.hip -o tools/micro/wide_probe && tools/micro/wide_probe [C] [iters]
#include <cstdio>
#include <vector>
#include "../../skyrim_amd/csrc/fused_block_wide.hip"
#include "../../skyrim_amd/csrc/fused_block2.hip"

using namespace skp;

template <class S>
static float run_wide(const Block2Args<f16>& a, int iters, int grid_override) {
    auto kern = proj_mlp_wide_kernel<f16, S>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, S::SMEM);
    const int ntiles = (a.M + S::BM - 1) / S::BM;
    const int grid = grid_override > 0 ? grid_override : 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(S::THREADS), S::SMEM, 0, a, ntiles);
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(S::THREADS), S::SMEM, 0, a, ntiles);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    if (hipGetLastError() != hipSuccess) printf("launch error\n");
    return ms / iters;
}

template <class S>
static float run_duo(const Block2Args<f16>& a, int iters) {
    auto kern = proj_mlp2_kernel<f16, S>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, S::SMEM);
    const unsigned grid = (unsigned)((a.M + S::BM - 1) / S::BM);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(S::THREADS), S::SMEM, 0, a);
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(S::THREADS), S::SMEM, 0, a);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms / iters;
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20;
    const int C = 384, M = 8 * 91 * 180;
    const long long plane = (long long)M * C;
    std::vector<f16> h((size_t)plane);
    unsigned r = 12345;
    for (auto& v : h) { r = r * 1664525u + 1013904223u; v = (f16)(((int)(r >> 20) - 2048) / 4096.0f); }
    f16 *ao, *xs, *pw, *w1, *w2;
    float* fb;
    int* winv;
    hipMalloc(&ao, 2 * plane * 2 + (144 * 16 * C * 4)); hipMalloc(&xs, 2 * plane * 2);
    hipMalloc(&pw, (size_t)C * C * 2); hipMalloc(&w1, (size_t)C * 4 * C * 2); hipMalloc(&w2, (size_t)C * 4 * C * 2);
    hipMalloc(&fb, (size_t)(8 * C + 4 * C) * 4); hipMalloc(&winv, (size_t)M * 4);
    hipMemcpy(ao, h.data(), plane * 2, hipMemcpyHostToDevice); hipMemset(ao + plane, 0, plane * 2);
    hipMemcpy(xs, h.data(), plane * 2, hipMemcpyHostToDevice); hipMemset(xs + plane, 0, plane * 2);
    std::vector<f16> hw((size_t)C * 4 * C);
    for (auto& v : hw) { r = r * 1664525u + 1013904223u; v = (f16)(((int)(r >> 20) - 2048) / 65536.0f); }
    hipMemcpy(pw, hw.data(), (size_t)C * C * 2, hipMemcpyHostToDevice);
    hipMemcpy(w1, hw.data(), hw.size() * 2, hipMemcpyHostToDevice); hipMemcpy(w2, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    std::vector<float> hb(12 * C, 0.01f);
    for (int i = C; i < 2 * C; ++i) hb[i] = 1.f;
    for (int i = 10 * C; i < 11 * C; ++i) hb[i] = 1.f;
    hipMemcpy(fb, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
    std::vector<int> hi(M);
    for (int i = 0; i < M; ++i) hi[i] = (int)((((long long)(i / 12) * 7919) % (M / 12)) * 12 + i % 12);   // runs of 12 consecutive rows, like the window table's
    hipMemcpy(winv, hi.data(), (size_t)M * 4, hipMemcpyHostToDevice);
    // proj_b, g1, e1, b1 (4C), b2, g2, e2
    Block2Args<f16> a{ao, plane, M, xs, plane, winv, pw, w1, w2, fb, fb + C, fb + 2 * C, fb + 3 * C, fb + 7 * C, fb + 10 * C, fb + 11 * C, 1e-5f};
    hipDeviceSynchronize();
#define W(name, ...) printf("%-44s %.4f ms\n", name, run_wide<WideShape<__VA_ARGS__>>(a, iters, 0)); fflush(stdout);
#define D(name, ...) printf("%-44s %.4f ms\n", name, run_duo<Blk2Shape<__VA_ARGS__>>(a, iters)); fflush(stdout);
    D("duo two-term", 384, 1, 2, false, true, 0, false)
    D("duo one-term", 384, 1, 2, false, true, 0, true)
    D("8-wave one-term", 384, 1, 2, false, false, 0, true)
    D("8-wave skew one-term", 384, 1, 2, true, false, 0, true)
    D("8-wave one-term rd3", 384, 1, 3, false, false, 0, true)
    D("8-wave one-term no-dma", 384, 1, 2, false, false, 4, true)
    D("8-wave one-term no-io", 384, 1, 2, false, false, 16, true)
    D("8-wave one-term no-gelu", 384, 1, 2, false, false, 1, true)
    D("8-wave one-term one-lds", 384, 1, 2, false, false, 2, true)
    D("8-wave one-term no-bar", 384, 1, 2, false, false, 8, true)
    D("8-wave one-term all off", 384, 1, 2, false, false, 31, true)
    D("8-wave two-term", 384, 1, 2, false, false, 0, false)
    return 0;
}
