/* C ABI of the gfx950 SFNO (FourCastNet v2-small) building blocks.
 *
 * Replaces, for the SFNO row of the north star, what the reference reaches through
 * earth2mip.networks.fcnv2_sm.load(...) (/root/reference/skyrim/core/models/fourcastnet_v2.py:36-37): the forward of
 * modulus' SphericalFourierNeuralOperatorNet on torch / torch-harmonics.  Every linear map of that network (1x1
 * convolutions, truncated real DFT, Legendre analysis / synthesis, per-degree complex channel mixing) is one call of
 * sksfno_gemm_run against a constant matrix prepared by sksfno_prepare_weight; sksfno_instance_norm is the only other
 * kernel.  The host side (skyrim_amd/sfno/engine.py) owns the buffers and the order of the calls.
 * All pointers are device pointers; calls are asynchronous on `stream` (a hipStream_t); nothing is allocated inside. */
#ifndef SKYRIM_SFNO_H
#define SKYRIM_SFNO_H

#ifdef __cplusplus
extern "C" {
#endif

#define SKSFNO_ABI_VERSION 1
#define SKSFNO_E_ARG (-1) /* bad argument */
#define SKSFNO_E_HIP (-2) /* a HIP call failed */

/* out[b](m, n) = act( sum_k A[b](m, k) W[b][n][k] + bias[n] + res_pre[b](m, n) ) + res_post[b](m, n),  b < batch.
 * A[b](m, k)  = a[b * a_sb + (m / a_m1) * a_sm2 + (m % a_m1) * a_sm + k * a_sk]      (fp32)
 * out[b](m,n) at out[b * o_sb + (m / o_m1) * o_sm2 + (m % o_m1) * o_sm + n * o_sn]   (fp32; res_pre / res_post likewise)
 * (batch b of a ragged launch: see k_lo_step / m_cap_step below)
 * W[b]        = fp16 hi plane [N][ldw] at w + b * w_sb (elements), lo plane at + w_plane (sksfno_prepare_weight). */
typedef struct sksfno_gemm {
    const float* a;
    long long a_sb;
    int a_m1;
    long long a_sm, a_sm2, a_sk;
    const void* w;
    long long w_sb, w_plane;
    int ldw;
    const float* bias;     /* [N] or NULL */
    const float* res_pre;  /* NULL or same addressing as out */
    const float* res_post; /* NULL or same addressing as out */
    float* out;
    long long o_sb;
    int o_m1;
    long long o_sm, o_sm2, o_sn;
    int M, N, K, batch;
    int act;               /* 0 = none, 1 = erf-GELU, 2 = swish */
    /* ragged batches (spherical harmonics are zero for l < m): batch b contracts only k >= (b * k_lo_step) rounded down to a
     * multiple of 32, and computes only rows m < m_cap0 + b * m_cap_step when m_cap_step > 0 (other rows are left untouched). */
    int k_lo_step, m_cap0, m_cap_step;
    /* optional per-k affine applied to A before it is split into fp16 hi/lo: A'(m, k) = A(m, k) * a_kscale[k] + a_kshift[k]
     * (input normalisation -- raw fields such as geopotential or pressure exceed the fp16 range); both or neither */
    const float* a_kscale;
    const float* a_kshift;
    /* optional second A source for k >= a2_k_split (concatenation along K, e.g. the big-skip concat of the decoder): same row
     * addressing as `a`, k stride a2_sk; a2_k_split a multiple of 8; batch == 1 */
    const float* a2;
    long long a2_sk;
    int a2_k_split;
    /* MFMA terms: 3 (or 0) = A and W as fp16 hi/lo pairs (fp32-class product); 2 = A rounded to one fp16 plane, W hi/lo */
    int terms;
} sksfno_gemm;

int sksfno_abi_version(void);

/* dst[n][k] (ld = ldw, a multiple of 8 >= K, zero-filled beyond K) = hi/lo fp16 split of src[n * sn + k * sk];
 * hi plane at dst, lo plane at dst + plane (elements, >= N * ldw). */
int sksfno_prepare_weight(const float* src, long long sn, long long sk, int N, int K, void* dst, long long plane, int ldw, void* stream);

int sksfno_gemm_run(const sksfno_gemm* desc, void* stream);

/* out[c][i] = (x[c][i] - mean_c) * rsqrt(var_c + eps) * gamma[c] + beta[c] over i < HW (biased variance), c < C */
int sksfno_instance_norm(const float* x, const float* gamma, const float* beta, float* out, int C, long long HW, float eps, void* stream);

#ifdef __cplusplus
}
#endif
#endif
