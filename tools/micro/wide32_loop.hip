// Feasibility probe for a ONE-wave-per-SIMD row-tile block kernel on v_mfma_f32_32x32x16_f16 (round 6): the instruction mix of the
// C = 384 one-term block (projection, fc1 -> GELU -> fc2 on 32 tokens per wave, weights streamed through a three-unit LDS ring by LDS-DMA)
// on synthetic buffers.  Results are NOT the block's (no LayerNorm algebra, fragment orders arbitrary): what is measured is whether a
// compiler-scheduled single wave keeps the matrix pipe busy with the LDS reads, DMA requests, barriers and GELU between its MFMAs.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/micro/wide32_loop.hip -o tools/micro/wide32_loop && tools/micro/wide32_loop
// PROBE bits: 1 no GELU, 2 no LDS reads (one fragment reused), 4 no DMA, 8 no barriers, 16 no row I/O, 32 GELU packed (v_pk_fma_f32)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    const unsigned dst = __builtin_amdgcn_readfirstlane(lds_dst);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}
__device__ __forceinline__ float gelu1(float x) {
    const float a = __builtin_amdgcn_fmed3f(__builtin_fabsf(x), 0.f, 5.9396970f);
    float p = __builtin_fmaf(a, -6.177511978e-07f, 1.091520153e-05f);
    p = __builtin_fmaf(a, p, -4.273382365e-05f);
    p = __builtin_fmaf(a, p, -4.982745158e-04f);
    p = __builtin_fmaf(a, p, 7.545167115e-03f);
    p = __builtin_fmaf(a, p, -5.282834917e-02f);
    p = __builtin_fmaf(a, p, -4.591012597e-01f);
    p = __builtin_fmaf(a, p, -1.151117682e+00f);
    p = p * a;
    float t = 1.0f - __builtin_amdgcn_exp2f(p);
    t = __builtin_copysignf(t, x);
    return (0.5f * x) * (1.0f + t);
}
__device__ __forceinline__ f32x2 gelu2(f32x2 x) {
    f32x2 a;
    a.x = __builtin_amdgcn_fmed3f(__builtin_fabsf(x.x), 0.f, 5.9396970f);
    a.y = __builtin_amdgcn_fmed3f(__builtin_fabsf(x.y), 0.f, 5.9396970f);
    f32x2 p = __builtin_elementwise_fma(a, (f32x2)(-6.177511978e-07f), (f32x2)(1.091520153e-05f));
    p = __builtin_elementwise_fma(a, p, (f32x2)(-4.273382365e-05f));
    p = __builtin_elementwise_fma(a, p, (f32x2)(-4.982745158e-04f));
    p = __builtin_elementwise_fma(a, p, (f32x2)(7.545167115e-03f));
    p = __builtin_elementwise_fma(a, p, (f32x2)(-5.282834917e-02f));
    p = __builtin_elementwise_fma(a, p, (f32x2)(-4.591012597e-01f));
    p = __builtin_elementwise_fma(a, p, (f32x2)(-1.151117682e+00f));
    p = p * a;
    f32x2 t;
    t.x = 1.0f - __builtin_amdgcn_exp2f(p.x);
    t.y = 1.0f - __builtin_amdgcn_exp2f(p.y);
    t.x = __builtin_copysignf(t.x, x.x);
    t.y = __builtin_copysignf(t.y, x.y);
    return ((f32x2)(0.5f) * x) * ((f32x2)(1.0f) + t);
}

constexpr int C = 384, KS = C / 16, NB = C / 32, NCH = 4 * C / 32, NWAVES = 4;
constexpr int SLOT = KS * 1024, UNIT = 2 * SLOT, NRING = 3, NPU = NB / 2, NU = NPU + 1 + NCH, HALF = KS / NWAVES, PIECES = 2 * HALF;
constexpr int SMEM = NRING * UNIT + 4 * C * 4;

template <int PROBE, int ORDER>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
k(const f16* __restrict__ wp, const f16* __restrict__ w1, const f16* __restrict__ w2, const f16* __restrict__ x, f16* __restrict__ out, const float* __restrict__ b1, int ntiles) {
    constexpr bool P_GELU = PROBE & 1, P_LDS = PROBE & 2, P_DMA = PROBE & 4, P_BAR = PROBE & 8, P_IO = PROBE & 16, P_PK = PROBE & 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* tab = reinterpret_cast<float*>(smem + NRING * UNIT);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds_base = (unsigned)(size_t)smem;
    const char* lrd = smem + lane * 16;

    auto dma_unit = [&](int u, int rp) {
        const f16 *s0, *s1;
        if (u < NPU) { s0 = wp + ((long long)(2 * u) * KS << 9); s1 = s0 + (KS << 9); }
        else {
            const int j = u - NPU - 1;
            s0 = w1 + ((long long)(j + 1 < NCH ? j + 1 : j) * KS << 9);
            s1 = w2 + ((long long)(j < 0 ? 0 : j) * KS << 9);
        }
        s0 += lane * 8 + (wave << 9); s1 += lane * 8 + (wave << 9);
        const unsigned dst = lds_base + (unsigned)(rp * UNIT) + (unsigned)(wave << 10);
#pragma unroll
        for (int i = 0; i < HALF; ++i) glds16(s0 + (i * NWAVES << 9), dst + (unsigned)(i * NWAVES << 10));
#pragma unroll
        for (int i = 0; i < HALF; ++i) glds16(s1 + (i * NWAVES << 9), dst + (unsigned)(SLOT + (i * NWAVES << 10)));
    };

    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    dma_unit(0, 0);
    dma_unit(1, 1);
    for (int i = tid; i < 4 * C; i += 256) tab[i] = b1[i];

    bool has_next = false;
    int rp = 0;
    auto top = [&](int u, bool pending) {
        const bool mlp = u > NPU;
        if (!(P_BAR && mlp)) {
            if (pending) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PIECES) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        const int rp2 = rp == 0 ? 2 : rp - 1;
        if (P_DMA && mlp && u + 2 < NU) return false;
        if (u + 2 < NU) { dma_unit(u + 2, rp2); return true; }
        if (has_next) { dma_unit(u + 2 - NU, rp2); return true; }
        return false;
    };
    auto advance = [&] { rp = rp == 2 ? 0 : rp + 1; };
    auto rdfrag = [&](const char* p) { return *reinterpret_cast<const f16x8*>(p); };

    for (; tile < ntiles; tile += gridDim.x) {
        has_next = tile + (int)gridDim.x < ntiles;
        const long long row0 = ((long long)tile * NWAVES + wave) * 32;
        f16x8 xh[KS];
        {
            const f16* p = x + (row0 + (lane & 31)) * C + (lane >> 5) * 8;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if constexpr (P_IO) { xh[ks] = f16x8{}; xh[ks][0] = (f16)(float)(lane + ks); }
                else xh[ks] = *reinterpret_cast<const f16x8*>(p + ks * 16);
            }
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) asm volatile("" : "+v"(xh[ks]));
        }
        f32x16 yacc[NB];
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
            for (int v = 0; v < 16; ++v) yacc[n][v] = 0.f;

        // ---- projection: two 32-column blocks per unit ---- //
        bool pending = false;
#pragma unroll
        for (int u = 0; u < NPU; ++u) {
            pending = top(u, pending);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const char* st = lrd + rp * UNIT + h * SLOT;
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) yacc[2 * u + h] = __builtin_amdgcn_mfma_f32_32x32x16_f16(rdfrag(st + (P_LDS ? 0 : ks << 10)), xh[ks], yacc[2 * u + h], 0, 0, 0);
            }
            advance();
        }
        // ---- fake LayerNorm: a row reduction + rewrite of the operand ---- //
        {
            float s = 0.f;
#pragma unroll
            for (int n = 0; n < NB; ++n)
#pragma unroll
                for (int v = 0; v < 16; ++v) s += yacc[n][v];
            s += __shfl_xor(s, 32);
            const float m = s * (1.f / C);
#pragma unroll
            for (int n = 0; n < NB; ++n) {
#pragma unroll
                for (int v = 0; v < 16; ++v) { xh[2 * n + (v >> 3)][v & 7] = (f16)((yacc[n][v] - m) * 0.01f); yacc[n][v] = 0.f; }
            }
        }
        // ---- MLP: interval j = MFMAs of fc1(j + 2) and fc2(j) in ORDER, GELU(j + 1) spliced between them ---- //
        // ORDER 0: fc1 (one chain) then fc2 (n outer, s inner: dependent pairs)    1: fc1 one chain, fc2 s outer
        //       2: fc1 two chains (even / odd k-steps), fc2 s outer                3: fc1 (one chain) and fc2 alternating
        //       4: fc1 two chains and fc2 alternating
        f32x16 hacc[2];
        float gv[16];
        f16x8 hh[2];
        auto zero_h = [&] {
#pragma unroll
            for (int v = 0; v < 16; ++v) { hacc[0][v] = 0.f; hacc[1][v] = 0.f; }
        };
        auto take = [&](int j) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 bb = *reinterpret_cast<const float4*>(tab + j * 32 + 8 * q + 4 * (lane >> 5));
                const int o = 4 * q;
                if constexpr (ORDER == 2 || ORDER == 4) {
                    gv[o] = hacc[0][o] + hacc[1][o] + bb.x; gv[o + 1] = hacc[0][o + 1] + hacc[1][o + 1] + bb.y;
                    gv[o + 2] = hacc[0][o + 2] + hacc[1][o + 2] + bb.z; gv[o + 3] = hacc[0][o + 3] + hacc[1][o + 3] + bb.w;
                } else {
                    gv[o] = hacc[0][o] + bb.x; gv[o + 1] = hacc[0][o + 1] + bb.y; gv[o + 2] = hacc[0][o + 2] + bb.z; gv[o + 3] = hacc[0][o + 3] + bb.w;
                }
            }
        };
        top(NPU, false);
        zero_h();
        {
            const char* st = lrd + rp * UNIT;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) hacc[ks & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(rdfrag(st + (P_LDS ? 0 : ks << 10)), xh[ks], hacc[ks & 1], 0, 0, 0);
        }
        take(0);
        hh[0] = xh[0]; hh[1] = xh[1];
        advance();
        pending = true;
        for (int j = 0; j < NCH; ++j) {
            pending = top(NPU + 1 + j, pending);
            const char* st = lrd + rp * UNIT;
            zero_h();
            f16x8 hn[2];
            auto f1 = [&](int ks) {
                constexpr int two = (ORDER == 2 || ORDER == 4);
                const int c = two ? (ks & 1) : 0;
                hacc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(rdfrag(st + (P_LDS ? 0 : ks << 10)), xh[ks], hacc[c], 0, 0, 0);
            };
            auto f2 = [&](int n, int s) {
                yacc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(rdfrag(st + SLOT + (P_LDS ? 0 : (2 * n + s) << 10)), hh[s], yacc[n], 0, 0, 0);
            };
            auto valu = [&](int i) {            // slot i of 48: GELU pieces first, conversions after
                if constexpr (!P_GELU) {
                    if constexpr (P_PK) { if (i < 8) { const f32x2 r = gelu2(f32x2{gv[2 * i], gv[2 * i + 1]}); gv[2 * i] = r.x; gv[2 * i + 1] = r.y; } }
                    else { if (i < 16) gv[i] = gelu1(gv[i]); }
                }
                if (i == 40 || i == 41) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) hn[i - 40][e] = (f16)gv[8 * (i - 40) + e];
                }
            };
            if constexpr (ORDER <= 2) {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) { f1(ks); valu(ks); }
                if constexpr (ORDER == 0) {
#pragma unroll
                    for (int n = 0; n < NB; ++n) { f2(n, 0); valu(24 + 2 * n); f2(n, 1); valu(25 + 2 * n); }
                } else {
#pragma unroll
                    for (int s = 0; s < 2; ++s)
#pragma unroll
                        for (int n = 0; n < NB; ++n) { f2(n, s); valu(24 + s * NB + n); }
                }
            } else {
#pragma unroll
                for (int i = 0; i < KS; ++i) { f1(i); valu(2 * i); f2(i % NB, i / NB); valu(2 * i + 1); }
            }
            hh[0] = hn[0]; hh[1] = hn[1];
            take(j + 1 < NCH ? j + 1 : j);
            advance();
        }
        // ---- epilogue: row reduction, hi / lo planes out ---- //
        {
            float s = 0.f;
#pragma unroll
            for (int n = 0; n < NB; ++n)
#pragma unroll
                for (int v = 0; v < 16; ++v) s += yacc[n][v];
            s += __shfl_xor(s, 32);
            const float m = s * (1.f / C);
            if (P_IO && m != 12345.f) continue;
            f16* dst = out + (row0 + (lane & 31)) * C + (lane >> 5) * 16;
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                f16x8 o0, o1, l0, l1;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float a = yacc[n][i] - m + (float)xh[2 * n][i], b = yacc[n][8 + i] - m + (float)xh[2 * n + 1][i];
                    o0[i] = (f16)a; o1[i] = (f16)b; l0[i] = (f16)(a - (float)o0[i]); l1[i] = (f16)(b - (float)o1[i]);
                }
                *reinterpret_cast<f16x8*>(dst + n * 32) = o0; *reinterpret_cast<f16x8*>(dst + n * 32 + 8) = o1;
                *reinterpret_cast<f16x8*>(dst + (long long)ntiles * 128 * C + n * 32) = l0; *reinterpret_cast<f16x8*>(dst + (long long)ntiles * 128 * C + n * 32 + 8) = l1;
            }
        }
    }
}

template <int PROBE, int ORDER>
static void run(const char* name, const f16* wp, const f16* w1, const f16* w2, const f16* x, f16* out, const float* b1, int ntiles, int iters) {
    auto kern = k<PROBE, ORDER>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(256), SMEM, 0, wp, w1, w2, x, out, b1, ntiles);
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(256), SMEM, 0, wp, w1, w2, x, out, b1, ntiles);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const hipError_t e = hipGetLastError();
    const double fl = 2.0 * ntiles * 128 * (384.0 * 384 + 2.0 * 384 * 1536);
    printf("%-40s %.4f ms  %.0f TF/s %s\n", name, ms / iters, fl / (ms / iters * 1e-3) / 1e12, e == hipSuccess ? "" : hipGetErrorString(e));
    fflush(stdout);
}

int main(int argc, char** argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20;
    const int M = 8 * 91 * 180, ntiles = (M + 127) / 128;
    const size_t n = (size_t)ntiles * 128 * C;
    std::vector<f16> h(n);
    unsigned r = 12345;
    for (auto& v : h) { r = r * 1664525u + 1013904223u; v = (f16)(((int)(r >> 20) - 2048) / 4096.0f); }
    f16 *x, *out, *wp, *w1, *w2;
    float* b1;
    hipMalloc(&x, n * 2); hipMalloc(&out, n * 4);
    hipMalloc(&wp, (size_t)C * C * 2); hipMalloc(&w1, (size_t)4 * C * C * 2); hipMalloc(&w2, (size_t)4 * C * C * 2); hipMalloc(&b1, 4 * C * 4);
    hipMemcpy(x, h.data(), n * 2, hipMemcpyHostToDevice);
    std::vector<f16> hw((size_t)4 * C * C);
    for (auto& v : hw) { r = r * 1664525u + 1013904223u; v = (f16)(((int)(r >> 20) - 2048) / 65536.0f); }
    hipMemcpy(wp, hw.data(), (size_t)C * C * 2, hipMemcpyHostToDevice);
    hipMemcpy(w1, hw.data(), hw.size() * 2, hipMemcpyHostToDevice); hipMemcpy(w2, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    std::vector<float> hb(4 * C, 0.01f);
    hipMemcpy(b1, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
    hipDeviceSynchronize();
#define R(P, O, name) run<P, O>(name, wp, w1, w2, x, out, b1, ntiles, iters);
    R(0, 0, "order0 full") R(0, 1, "order1 full") R(0, 2, "order2 full") R(0, 3, "order3 full") R(0, 4, "order4 full")
    R(31, 0, "order0 all off") R(31, 1, "order1 all off") R(31, 2, "order2 all off") R(31, 3, "order3 all off") R(31, 4, "order4 all off")
    R(32, 4, "order4 packed GELU") R(1, 4, "order4 no-gelu") R(2, 4, "order4 one-lds") R(4, 4, "order4 no-dma") R(8, 4, "order4 no-bar") R(16, 4, "order4 no-io")
    return 0;
}
