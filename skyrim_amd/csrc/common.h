// Shared device-side vocabulary for the gfx950 Pangu engine: 16-bit operand types,
// MFMA wrappers, hi/lo operand splitting and the LDS swizzle.  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace skp {

typedef _Float16 f16;
typedef __bf16 bf16;
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef bf16 bf16x8 __attribute__((ext_vector_type(8)));

// ---- operand type traits ------------------------------------------------- //
template <class T> struct OpT;
template <> struct OpT<f16> {
    typedef f16x8 v8;
    // D(16x16) += A(16x32) * B(32x16); lane l holds A[l&15][8*(l>>4)+j], B[8*(l>>4)+j][l&15];
    // D: col = l&15, row = 4*(l>>4)+r.
    static __device__ __forceinline__ f32x4 mfma(v8 a, v8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};
template <> struct OpT<bf16> {
    typedef bf16x8 v8;
    static __device__ __forceinline__ f32x4 mfma(v8 a, v8 b, f32x4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};

template <class T>
__device__ __forceinline__ typename OpT<T>::v8 as_v8(const uint4& u) {
    return __builtin_bit_cast(typename OpT<T>::v8, u);
}

// ---- precision modes ------------------------------------------------------ //
// A GEMM computes sum over "terms": (A_hi,W_hi) [+ (A_lo,W_hi) if NA==2] [+ (A_hi,W_lo) if NW==2].
// bf16x3 carries 16 significand bits per operand: fp32-class results on the bf16 MFMA pipe.
struct PrecBF16x3 { typedef bf16 T; static constexpr int NA = 2, NW = 2; };
// fp16 hi/lo planes, 3 terms: 22 significand bits per operand (bf16x3: 16); needs |x| < 65504 (activations are O(1..10))
struct PrecF16x3  { typedef f16 T;  static constexpr int NA = 2, NW = 2; };
// fc2 of the "fp16 hidden" mode: A = single fp16 plane (the GELU output), W = fp16 hi/lo planes, 2 MFMA terms
struct PrecF16x2W { typedef f16 T;  static constexpr int NA = 1, NW = 2; };

// ---- hi/lo split ---------------------------------------------------------- //
// 8 fp32 -> NP planes of 8 x T packed as uint4.  lo = T(v - float(hi)).
// The hi plane is made OPAQUE (an empty asm that "modifies" it) before the lo plane is derived from it.  Left alone, hipcc converts twice --
// v_cvt_pk_f16_f32 for the plane that is stored, v_cvt_f16_f32 for the value the remainder is taken from -- and on gfx950 the two do not
// always agree: measured in round 4 (graphcast_fused.hip) as an error of one full fp16 ulp in single elements, ~1e-5 of all elements, always
// where the fp32 value sits within one fp32 ulp of an fp16 grid point.  With the barrier the remainder belongs to the bits that are stored.
template <class T, int NP>
__device__ __forceinline__ void split8(const float (&v)[8], uint4 (&out)[NP]) {
    typename OpT<T>::v8 h;
#pragma unroll
    for (int i = 0; i < 8; ++i) h[i] = (T)v[i];
    if constexpr (NP == 2) asm volatile("" : "+v"(h));
    out[0] = __builtin_bit_cast(uint4, h);
    if constexpr (NP == 2) {
        typename OpT<T>::v8 l;
#pragma unroll
        for (int i = 0; i < 8; ++i) l[i] = (T)(v[i] - (float)h[i]);
        out[1] = __builtin_bit_cast(uint4, l);
    }
}

template <class T, int NP>
__device__ __forceinline__ void split4(const float (&v)[4], uint2 (&out)[NP]) {
    typedef T t4 __attribute__((ext_vector_type(4)));
    t4 h;
#pragma unroll
    for (int i = 0; i < 4; ++i) h[i] = (T)v[i];
    if constexpr (NP == 2) asm volatile("" : "+v"(h));
    out[0] = __builtin_bit_cast(uint2, h);
    if constexpr (NP == 2) {
        t4 l;
#pragma unroll
        for (int i = 0; i < 4; ++i) l[i] = (T)(v[i] - (float)h[i]);
        out[1] = __builtin_bit_cast(uint2, l);
    }
}

// store 4 consecutive activations (fp32 or T)
template <class S>
__device__ __forceinline__ void store4(S* p, const float (&v)[4]) {
    if constexpr (sizeof(S) == 4) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
        uint2 o[1];
        split4<S, 1>(v, o);
        *reinterpret_cast<uint2*>(p) = o[0];
    }
}

// Blocked operand layout of everything the DMA GEMMs read: a [R][K] matrix (R % 16 == 0, K % 32 == 0) is stored as
// [R/16][K/32] blocks of 16 rows x 32 columns (1 KiB of 16-bit elements), so that one LDS-DMA instruction
// (64 lanes x 16 B) reads ONE contiguous, fully used 1 KiB block -- measured 43 B/clk/CU from L2 vs 21-24 B/clk/CU
// for the same bytes fetched as 16 rows x 64 B out of a row-major matrix (tools/micro/dma_bw.hip).
__device__ __host__ __forceinline__ long long blk_off(long long row, int col, int K) {
    return (((row >> 4) * (K >> 5) + (col >> 5)) << 9) + ((row & 15) << 5) + (col & 31);
}

// store 4 consecutive values as NP 16-bit planes (hi at p, lo at p + plane)
template <class T, int NP>
__device__ __forceinline__ void store4_planes(T* p, long long plane, const float (&v)[4]) {
    uint2 o[NP];
    split4<T, NP>(v, o);
    *reinterpret_cast<uint2*>(p) = o[0];
    if constexpr (NP == 2) *reinterpret_cast<uint2*>(p + plane) = o[1];
}

// store 8 consecutive values as NP 16-bit planes: one 16-byte store per plane
template <class T, int NP>
__device__ __forceinline__ void store8_planes(T* p, long long plane, const float (&v)[8]) {
    uint4 o[NP];
    split8<T, NP>(v, o);
    *reinterpret_cast<uint4*>(p) = o[0];
    if constexpr (NP == 2) *reinterpret_cast<uint4*>(p + plane) = o[1];
}

// "perm8" row order of the prepared weights: inside every aligned group of 32 output columns, prepared row
// rho = 16 jj + 4 q + i holds output column 8 q + 4 jj + i.  In the swapped MFMA order (lane: row l&15, columns
// 16 b + 4 (l>>4) + i of fragment b) the accumulators of a fragment pair (2bp, 2bp+1) are then 8 CONSECUTIVE output
// columns  n0w + 32 bp + 8 (l>>4) + [0..7]:  every activation store / residual load is 16 bytes per lane and one wave
// instruction covers one whole contiguous 1 KiB block of the blocked layout (8-byte stores are issue-bound at
// ~7 B/clk/CU, MI355X_MICROARCH.md "attention epilogue store tail").
__device__ __host__ __forceinline__ int perm8_col(int rho) {
    const int r = rho & 31;
    return (rho & ~31) + 8 * ((r >> 2) & 3) + 4 * (r >> 4) + (r & 3);
}

// ---- LDS tile addressing --------------------------------------------------- //
// A [rows][BK] tile of 16-bit elements is stored as 16-byte slots; slot s of row r lives at
// physical slot s ^ f(r) so that the 16-lane groups of ds_read_b128 (rows l&15, k-group l>>4)
// and the 8-lane groups of ds_write_b128 hit 16 / 8 distinct bank quads.
template <int BK>
__device__ __forceinline__ int lds_off(int row, int slot) {
    if constexpr (BK == 64) return row * 128 + ((slot ^ (row & 7)) << 4);
    else                    return row * 64 + ((slot ^ ((row >> 1) & 3)) << 4);
}

// erf-GELU = 0.5 x erfc(-x/sqrt2) with erfc(u/sqrt2) = 2^Q(u) for u >= 0 (Q: degree-8 fit of log2 erfc, tools/gelu_fit.py)
// and erfc(-z) = 2 - erfc(z).  Max abs error 3.8e-7 (fp32 round-off class; libm-erf GELU differs by the same amount),
// ONE transcendental (v_exp_f32) and 8 FMAs: ~11 VALU issue slots per element where A&S 7.1.26 (rcp + exp) took ~20 and
// libm erff ~50 -- the fc1 epilogue evaluates 2e8 of these per launch and is VALU-bound.
typedef float f32x2 __attribute__((ext_vector_type(2)));
// two elements at a time so that the polynomial runs on v_pk_fma_f32 (coefficients splat in registers)
__device__ __forceinline__ f32x2 gelu_erf2(f32x2 x) {
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
    f32x2 t;                                  // erf(|x|/sqrt2) = 1 - erfc
    t.x = 1.0f - __builtin_amdgcn_exp2f(p.x);
    t.y = 1.0f - __builtin_amdgcn_exp2f(p.y);
    t.x = __builtin_copysignf(t.x, x.x);
    t.y = __builtin_copysignf(t.y, x.y);
    return ((f32x2)(0.5f) * x) * ((f32x2)(1.0f) + t);
}
__device__ __forceinline__ float gelu_erf(float x) { return gelu_erf2((f32x2){x, x}).x; }
// The same polynomial in Estrin form: 4 + 2 + 1 FMAs in three levels instead of a chain of 7 dependent ones.  Where the evaluation is
// spliced between MFMAs (fused_block2.hip) the dependent v_pk_fma chain of the Horner form costs a wait state per link; the Estrin
// form leaves the scheduler independent work at every level.  Same coefficients; the results differ by fp32 rounding only.
__device__ __forceinline__ f32x2 gelu_erf2e(f32x2 x) {
    f32x2 a;
    a.x = __builtin_amdgcn_fmed3f(__builtin_fabsf(x.x), 0.f, 5.9396970f);
    a.y = __builtin_amdgcn_fmed3f(__builtin_fabsf(x.y), 0.f, 5.9396970f);
    // Q(a) = a (c0 + c1 a + ... + c7 a^7), c0 = -1.151117682, ..., c7 = -6.177511978e-07
    const f32x2 a2 = a * a;
    const f32x2 q01 = __builtin_elementwise_fma(a, (f32x2)(-4.591012597e-01f), (f32x2)(-1.151117682e+00f));
    const f32x2 q23 = __builtin_elementwise_fma(a, (f32x2)(7.545167115e-03f), (f32x2)(-5.282834917e-02f));
    const f32x2 q45 = __builtin_elementwise_fma(a, (f32x2)(-4.273382365e-05f), (f32x2)(-4.982745158e-04f));
    const f32x2 q67 = __builtin_elementwise_fma(a, (f32x2)(-6.177511978e-07f), (f32x2)(1.091520153e-05f));
    const f32x2 a4 = a2 * a2;
    const f32x2 q03 = __builtin_elementwise_fma(a2, q23, q01);
    const f32x2 q47 = __builtin_elementwise_fma(a2, q67, q45);
    f32x2 p = __builtin_elementwise_fma(a4, q47, q03);
    p = p * a;
    f32x2 t;                                  // erf(|x|/sqrt2) = 1 - erfc
    t.x = 1.0f - __builtin_amdgcn_exp2f(p.x);
    t.y = 1.0f - __builtin_amdgcn_exp2f(p.y);
    t.x = __builtin_copysignf(t.x, x.x);
    t.y = __builtin_copysignf(t.y, x.y);
    return ((f32x2)(0.5f) * x) * ((f32x2)(1.0f) + t);
}
// GELU of the 8 consecutive columns held in a fragment pair
__device__ __forceinline__ void gelu_erf8(const f32x4& lo, const f32x4& hi, float (&v)[8]) {
    const f32x2 r0 = gelu_erf2((f32x2){lo[0], lo[1]}), r1 = gelu_erf2((f32x2){lo[2], lo[3]});
    const f32x2 r2 = gelu_erf2((f32x2){hi[0], hi[1]}), r3 = gelu_erf2((f32x2){hi[2], hi[3]});
    v[0] = r0.x; v[1] = r0.y; v[2] = r1.x; v[3] = r1.y; v[4] = r2.x; v[5] = r2.y; v[6] = r3.x; v[7] = r3.y;
}

constexpr int WIN_TOKENS = 144;
constexpr int HEAD_DIM = 32;

}  // namespace skp
