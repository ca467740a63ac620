// Timing + checksum harness for the row-tile block kernel (csrc/fused_block2.hip) on synthetic buffers, no Python:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I skyrim_amd/csrc [-DBLOCK_SRC='"path/to/other/fused_block2.hip"'] tools/micro/block_probe.hip -o tools/micro/block_probe
//   tools/micro/block_probe [iters]
// Prints ms per launch and a bit-level checksum of the stream after ONE launch on a fixed input (two builds of the same arithmetic agree).
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
#ifndef BLOCK_SRC
#define BLOCK_SRC "../../skyrim_amd/csrc/fused_block2.hip"
#endif
#include BLOCK_SRC

using namespace skp;

template <class S>
static void run(const char* name, int C, int M, int iters) {
    const long long plane = (long long)M * C;
    std::vector<f16> h((size_t)plane);
    unsigned r = 12345;
    for (auto& v : h) { r = r * 1664525u + 1013904223u; v = (f16)(((int)(r >> 20) - 2048) / 1024.0f); }
    f16 *ao, *xs, *xs0, *pw, *w1, *w2;
    float* fb;
    int* winv;
    hipMalloc(&ao, 2 * plane * 2 + (144 * 16 * C * 4)); hipMalloc(&xs, 2 * plane * 2); hipMalloc(&xs0, 2 * plane * 2);
    hipMalloc(&pw, (size_t)C * C * 2); hipMalloc(&w1, (size_t)C * 4 * C * 2); hipMalloc(&w2, (size_t)C * 4 * C * 2);
    hipMalloc(&fb, (size_t)(12 * C) * 4); hipMalloc(&winv, (size_t)M * 4);
    hipMemcpy(ao, h.data(), plane * 2, hipMemcpyHostToDevice);
    for (auto& v : h) { r = r * 1664525u + 1013904223u; v = (f16)(((int)(r >> 20) - 2048) / 4194304.0f); }
    hipMemcpy(ao + plane, h.data(), plane * 2, hipMemcpyHostToDevice);
    for (auto& v : h) { r = r * 1664525u + 1013904223u; v = (f16)(((int)(r >> 20) - 2048) / 512.0f); }
    hipMemcpy(xs0, h.data(), plane * 2, hipMemcpyHostToDevice); hipMemset(xs0 + plane, 0, plane * 2);
    std::vector<f16> hw((size_t)C * 4 * C);
    for (auto& v : hw) { r = r * 1664525u + 1013904223u; v = (f16)(((int)(r >> 20) - 2048) / 32768.0f); }
    hipMemcpy(pw, hw.data(), (size_t)C * C * 2, hipMemcpyHostToDevice);
    for (auto& v : hw) { r = r * 1664525u + 1013904223u; v = (f16)(((int)(r >> 20) - 2048) / 32768.0f); }
    hipMemcpy(w1, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    for (auto& v : hw) { r = r * 1664525u + 1013904223u; v = (f16)(((int)(r >> 20) - 2048) / 65536.0f); }
    hipMemcpy(w2, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    std::vector<float> hb(12 * C);
    for (auto& v : hb) { r = r * 1664525u + 1013904223u; v = ((int)(r >> 20) - 2048) / 20480.0f; }
    for (int i = C; i < 2 * C; ++i) hb[i] += 1.f;                 // g1
    for (int i = 8 * C; i < 9 * C; ++i) hb[i] += 1.f;             // g2
    hipMemcpy(fb, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
    std::vector<int> hi(M);
    for (int i = 0; i < M; ++i) hi[i] = (int)((((long long)(i / 12) * 7919) % (M / 12)) * 12 + i % 12);   // runs of 12 consecutive rows, like the window table's
    hipMemcpy(winv, hi.data(), (size_t)M * 4, hipMemcpyHostToDevice);
    // proj_b, g1, e1, b1 (4C), b2, g2, e2
    Block2Args<f16> a{ao, plane, M, xs, plane, winv, pw, w1, w2, fb, fb + C, fb + 2 * C, fb + 3 * C, fb + 7 * C, fb + 8 * C, fb + 9 * C, 1e-5f};
    hipMemcpy(xs, xs0, 2 * plane * 2, hipMemcpyDeviceToDevice);
    hipError_t e = launch_blk2<f16, S>(a, 0);
    hipDeviceSynchronize();
    std::vector<uint16_t> o((size_t)2 * plane);
    hipMemcpy(o.data(), xs, 2 * plane * 2, hipMemcpyDeviceToHost);
    uint64_t hsh = 1469598103934665603ull;
    double sum = 0;
    int bad = 0;
    for (size_t i = 0; i < o.size(); ++i) {
        hsh = (hsh ^ o[i]) * 1099511628211ull;
        if (i < (size_t)plane) { f16 v; memcpy(&v, &o[i], 2); const float f = (float)v; if (!(f == f) || f > 1e4f || f < -1e4f) ++bad; else sum += f; }
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) launch_blk2<f16, S>(a, 0);
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) launch_blk2<f16, S>(a, 0);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    printf("%-28s %.4f ms   checksum %016llx  sum %.6e  bad %d  %s\n", name, ms / iters, (unsigned long long)hsh, sum, bad, e == hipSuccess ? "" : hipGetErrorString(e));
    fflush(stdout);
    hipFree(ao); hipFree(xs); hipFree(xs0); hipFree(pw); hipFree(w1); hipFree(w2); hipFree(fb); hipFree(winv);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20;
    run<Blk2Shape<384, 1, true>>("C=384 one-term", 384, 8 * 91 * 180, iters);
    run<Blk2Shape<384, 1, false>>("C=384 two-term", 384, 8 * 91 * 180, iters);
    run<Blk2Shape<192, 2, false>>("C=192 two-term", 192, 8 * 181 * 360, iters);
    run<Blk2Shape<192, 2, true>>("C=192 one-term", 192, 8 * 181 * 360, iters);
    return 0;
}
