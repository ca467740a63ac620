// Host-side declarations shared by the engine's translation units: geometry, prepared-weight
// views, scratch views and the per-stage launchers (one 6-h Pangu step = a fixed sequence of these).
#pragma once
#include <hip/hip_runtime.h>
#include "common.h"

namespace skp {

struct Geom {
    int n_lat, n_lon, n_levels, n_channels, surf0;   // 721, 1440, 13, 69, 65
    int lat_top;                                      // input latitude front pad (to a multiple of 4)
    int Z, H1, W1, H2, W2;
    int Hp[2], top[2], nH[2], nW[2], types[2], nwin[2];   // per resolution (0: layers 1/4, 1: layers 2/3)
    int ntok[2], mwin[2];                                 // tokens, padded window tokens (nwin*144)
    int roll_sign;                                        // shifted-window blocks roll by roll_sign * (1, 3, 6) first (-1: Swin, +1: pseudocode as written)
    float mask_value;                                     // additive shifted-window mask (-100)
    int surface_last, qkv_order, bias_transposed;         // prepare-time conventions (skyrim_pangu.h)
};

template <class T>
struct LinW { const T* w; long long plane; int ldw; };   // [N][ldw] hi plane (+ lo plane at +plane)

template <class T>
struct BlockW {
    LinW<T> qkv, proj, fc1, fc2;
    const T *w1f, *w2f;      // fc1 / fc2 in the fused MLP's fragment order (fused_mlp.hip), or null
    const T *projf, *qkvf;   // proj / qkv in the row-tile kernels' fragment order (rowtile.hip), or null
    const T *projh, *w1h, *w2h;   // proj / fc1 / fc2 in fragment order, HI PLANE ONLY: the two-term block kernel (fused_block2.hip), or null
    const T* qkvh;                // qkv in fragment order, hi plane only: the one-term QKV of the term plan, or null
    const float *qkv_b, *proj_b, *fc1_b, *fc2_b, *n1_g, *n1_b, *n2_g, *n2_b;
    const f16* bias_exp;     // [types][heads][9][9][64][4] (+ shifted-window mask on odd blocks): the first attention kernel; or null
    const f16* bias_cmp;     // [types][heads][144][24] compact, w reversed (+ mask): the second attention kernel; or null
};

template <class T>
struct ModelW {
    const float *mean, *std, *istd, *masks;
    LinW<T> embed_u, embed_s;
    const float *embed_u_b, *embed_s_b;
    BlockW<T> blk[16];
    const float *down_g, *down_b;
    LinW<T> down;
    LinW<T> up1, up2;
    const float *up_g, *up_b;
    LinW<T> rec_u, rec_s;
    const float *rec_u_b, *rec_s_b;
    const int* widx[2][2];   // [resolution][roll] window gather table: mwin entries, -1 = padding
    const int* winv[2][2];   // its inverse: stream token -> window row (ntok entries)
};

template <class P>
struct Work {
    typedef typename P::T T;
    T *X1s, *X2s, *X4s;      // residual streams: hi/lo planes, blocked layout (always two planes); lo at + xs_plane[res]
    long long xs_plane[2];
    f16 *q, *k, *vt;         // single fp16 planes in every mode (attention.hip)
    long long qkv_plane;
    T *ao, *hid, *u;         // hi/lo planes: + ao_plane / hid_plane / u_plane
    long long ao_plane, hid_plane, u_plane;
    const T* zrow;           // zeros (padding rows of the DMA GEMMs)
    int qkv_a1;              // 1 (fp16 planes only): QKV reads only the hi plane of the stream (2 MFMA terms)
    float2* stats;
};

template <class P>
struct AttnArgs {
    const f16 *q, *k, *vt;
    long long plane;
    const f16* bias_exp;
    const f16* bias_cmp;     // non-null: earth_attention2_kernel (bias gathered from the compact table staged in LDS)
    typename P::T* out;
    long long out_plane;
    int ld_out, n_win, nW, heads;
    int out_planes = 0;      // 1: only the hi plane of the output is written (its reader is a one-term block kernel); 0: the mode's own
};

// layer index 0..3 -> resolution 0/1, channels, heads
inline int layer_res(int layer) { return (layer == 0 || layer == 3) ? 0 : 1; }
inline int layer_dim(int layer) { return layer_res(layer) == 0 ? 192 : 384; }
inline int layer_heads(int layer) { return layer_res(layer) == 0 ? 6 : 12; }

template <class P> hipError_t launch_attention(const AttnArgs<P>&, hipStream_t);

// A residual stream is one buffer of hi/lo planes (no fp32 copy): read by the next GEMM's DMA, read-modify-written by the
// LayerNorm epilogues.
template <class P> hipError_t op_embed(const Geom&, const ModelW<typename P::T>&, const float* state, typename P::T* X1s, const Work<P>&, hipStream_t);
template <class P> hipError_t op_recover(const Geom&, const ModelW<typename P::T>&, const typename P::T* skip_s, const typename P::T* x4_s, float* state, const Work<P>&, hipStream_t);
template <class P> hipError_t op_qkv(const Geom&, const BlockW<typename P::T>&, const int* widx, int res, const typename P::T* Xs, const Work<P>&, hipStream_t);
template <class P> hipError_t op_proj(const Geom&, const BlockW<typename P::T>&, const int* widx, int res, typename P::T* Xs, const Work<P>&, hipStream_t);
template <class P> hipError_t op_fc1(const Geom&, const BlockW<typename P::T>&, int res, const typename P::T* Xs, const Work<P>&, hipStream_t);
template <class P> hipError_t op_fc2(const Geom&, const BlockW<typename P::T>&, int res, typename P::T* Xs, const Work<P>&, hipStream_t);
template <class P> hipError_t op_down(const Geom&, const ModelW<typename P::T>&, const typename P::T* X1s, typename P::T* X2s, const Work<P>&, hipStream_t);
template <class P> hipError_t op_up(const Geom&, const ModelW<typename P::T>&, const typename P::T* X2s, typename P::T* X4s, const Work<P>&, hipStream_t);
// fc1 -> GELU -> fc2 -> LayerNorm -> residual in one kernel (3-term modes); weights from prep_mlp_weights
template <class P> hipError_t op_mlp_fused(const Geom&, const BlockW<typename P::T>&, int res, typename P::T* Xs, const Work<P>&, hipStream_t);
template <class T> hipError_t prep_mlp_weights(const float* w1, const float* w2, T* w1f, T* w2f, int C, hipStream_t, int planes = 2);
// row-tile forms of proj (+ LayerNorm + window reverse + residual) and of the 2-term QKV linear (rowtile.hip)
template <class P> hipError_t op_proj_rowtile(const Geom&, const BlockW<typename P::T>&, const int* widx, int res, typename P::T* Xs, const Work<P>&, hipStream_t);
hipError_t op_qkv_rowtile(const Geom&, const BlockW<f16>&, const int* widx, int res, const f16* Xs, const Work<PrecF16x3>&, hipStream_t);
template <class T> hipError_t prep_rowtile_weights(const float* w, T* wf, int N, int K, hipStream_t, int planes = 2);
// the 2-term / 1-term QKV linear and the window attention as ONE kernel: q / k / v stay in registers (attention.hip); out_planes as AttnArgs'
hipError_t op_qkv_attention(const Geom&, const BlockW<f16>&, const int* widx, int res, const f16* Xs, const Work<PrecF16x3>&, hipStream_t, int out_planes);
// proj + LayerNorm + residual + MLP + LayerNorm + residual in one kernel (fused_block.hip): weights from prep_rowtile_weights / prep_mlp_weights
template <class P> hipError_t op_proj_mlp_fused(const Geom&, const BlockW<typename P::T>&, const int* winv, int res, typename P::T* Xs, const Work<P>&, hipStream_t);
// the same with the weights as ONE fp16 plane: two MFMA terms, or -- `one` -- a single term (activation operands as one fp16 plane too) (fused_block2.hip)
hipError_t op_proj_mlp2(const Geom&, const BlockW<f16>&, const int* winv, int res, f16* Xs, const Work<PrecF16x3>&, hipStream_t, int one = 0);
template <class T, int NPL> hipError_t split_planes(const float* x, T* planes, long long plane, long long n, int C, hipStream_t);
template <class T> hipError_t merge_planes(const T* planes, long long plane, float* x, long long n, int C, hipStream_t);

// prepare-time helpers (aux.hip)
template <class T, int NW>
hipError_t prep_weight(const float* src, T* dst, long long plane, int N, int K, int ldd, long long sn, long long sk, int blocked, int perm, hipStream_t);
hipError_t prep_bias_expand(const float* table, f16* out, int types, int heads, int nH, int roll, float mask_value, hipStream_t, int transposed = 0);   // roll: 0 | -1 | +1
hipError_t prep_bias_compact(const float* table, f16* out, int types, int heads, int nH, int roll, float mask_value, hipStream_t, int transposed = 0);
hipError_t prep_qkv_rows(const float* w, const float* b, float* w_out, float* b_out, int C, int heads, hipStream_t);   // (heads, 3, hd) -> (3, heads, hd) rows
hipError_t prep_window_index(int* idx, int Z, int H, int W, int Hp, int top, int roll, hipStream_t, int surface_last = 0);
hipError_t prep_window_inverse(const int* idx, int n, int* inv, hipStream_t);
hipError_t prep_reciprocal(const float* src, float* dst, int n, hipStream_t);
// calibration of a two-term plan: column sums of an operand in plane layout; the dropped weight residue times their mean folded into a bias
template <class T> hipError_t colsum_planes(const T* x, long long plane, int nplanes, const int* rowmap, int rows, int K, float* scratch, float* out, hipStream_t);
size_t colsum_scratch_floats(int rows, int K);
hipError_t bias_fold(const float* w, const float* colsum, float scale, float* bias, int N, int K, hipStream_t);
template <class T> hipError_t merge_stats(const T* x, long long plane, float2* stats, int Z, int H1, int W1, int H2, int W2, int C, float eps, hipStream_t);

}  // namespace skp
