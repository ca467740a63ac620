/* C ABI of the gfx950 SFNO (FourCastNet v2-small) building blocks.
 *
 * Replaces, for the SFNO row of the north star, what the reference reaches through
 * earth2mip.networks.fcnv2_sm.load(...) (/root/reference/skyrim/core/models/fourcastnet_v2.py:36-37): the forward of
 * modulus' SphericalFourierNeuralOperatorNet on torch / torch-harmonics.  Every linear map of that network (1x1
 * convolutions, truncated real DFT, Legendre analysis / synthesis, per-degree complex channel mixing) is one call of
 * sksfno_gemm_run against a constant matrix prepared by sksfno_prepare_weight; sksfno_instance_norm normalises a field.
 * ABI v2 adds the fused pixel-wise chains (sksfno_chain_run: encoder, block MLP, last MLP + decoder as one kernel each, the
 * instance norm in front of an MLP as a per-channel affine from sksfno_instance_stats).
 * The host side (skyrim_amd/sfno/engine.py) owns the buffers and the order of the calls.
 * All pointers are device pointers; calls are asynchronous on `stream` (a hipStream_t); nothing is allocated inside. */
#ifndef SKYRIM_SFNO_H
#define SKYRIM_SFNO_H

#ifdef __cplusplus
extern "C" {
#endif

#define SKSFNO_ABI_VERSION 2
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

/* ---- fused pixel-wise chains (ABI v2) -------------------------------------------------------------------------------------
 * Activations are [channels][HW] fp32, HW a multiple of 16.  One launch walks the pixels once:
 *   ENC   out = W2 GELU(W1 (x * scale + shift) + b1) + res                 x: [KX][HW] raw state, res: position embedding [C][HW]
 *   MLP   out = W2 GELU(W1 (y * scale + shift) + b1) + b2 + res            y, res, out: [C][HW]   (scale / shift: sksfno_instance_stats)
 *   TAIL  z = MLP(y);  out = V2 GELU(V1 concat(z, x * xscale + xshift) + d1) + d2       out: [OUT][HW]
 * replacing, in /root/reference's network (fourcastnet_v2.py:36-37 -> modulus SphericalFourierNeuralOperatorNet), encoder,
 * block MLP (+ norm1) and decoder (+ big skip).  Weights are prepared by sksfno_prepare_chain_weights from matrices zero-padded to
 * the widths of the shape class (sksfno_chain_dims); `tab` is one fp32 array in the order
 *   scale[K0] shift[K0] b1[H0] b2[CP]  (TAIL: + xscale[KXP] xshift[KXP] d1[CP] d2[OP])      K0 = KXP, H0 = CP for ENC; K0 = CP, H0 = HP else,
 * zero beyond the real widths. */
#define SKSFNO_CHAIN_ENC 0
#define SKSFNO_CHAIN_MLP 1
#define SKSFNO_CHAIN_TAIL 2

typedef struct sksfno_chain {
    int mode;              /* SKSFNO_CHAIN_* */
    int shape;             /* shape class: 0 = (CP 256, HP 512, KXP 96, OP 96), 1 = (64, 96, 32, 32) */
    const float* y;        /* input of the first pair (ENC: the raw state) */
    const float* x;        /* TAIL: the raw state */
    const float* res;
    float* out;            /* may alias y (MLP) or x (TAIL): a workgroup reads its own pixels before it writes them */
    long long HW;
    int C, KX, OUT;        /* real widths: embed, state channels, output channels */
    const void *w1f, *w2f; /* first pair */
    const void *v1f, *v2f; /* TAIL: decoder pair */
    const float* tab;
} sksfno_chain;

/* padded widths of a shape class */
int sksfno_chain_dims(int shape, int* cp, int* hp, int* kxp, int* op);

/* w1: [H][K], w2: [N][H] fp32, zero-padded (K, H, N multiples of 32) -> fragment-order fp16 hi/lo planes; w1f: 2 H K, w2f: 2 N H elements */
int sksfno_prepare_chain_weights(const float* w1, const float* w2, int K, int H, int N, void* w1f, void* w2f, void* stream);

/* scale[c] = gamma[c] rstd_c, shift[c] = beta[c] - mean_c scale[c]: the instance norm of x[c][HW] as an affine of its consumer */
int sksfno_instance_stats(const float* x, const float* gamma, const float* beta, float* scale, float* shift, int C, long long HW, float eps, void* stream);

int sksfno_chain_run(const sksfno_chain* desc, void* stream);

#ifdef __cplusplus
}
#endif
#endif
