/* C ABI of the gfx950 GraphCast building blocks.
 *
 * Replaces, for the GraphCast row of the north star, what the reference reaches through
 * earth2mip.networks.graphcast.load_time_loop_operational(...) and stepper.step()
 * (/root/reference/skyrim/core/models/graphcast.py:51-54, 102-118): DeepMind's JAX typed graph network.  A step is a
 * sequence of interaction-network MLPs; each is  gather + concat -> Linear -> swish -> Linear -> LayerNorm (+ residual),
 * with a sum over incoming edges between the edge and the node update:
 *     skgc_gather_gemm   first Linear of an MLP on rows assembled from up to three row-major sources through index arrays
 *                        (edge latent | sender node | receiver node), bias + swish fused
 *     sksfno_gemm_run    (include/skyrim_sfno.h) second Linear, and the output layer that writes the next state
 *     skgc_layer_norm    LayerNorm over the latent (+ residual)
 *     skgc_segment_sum   sum of edge rows per receiver (edges sorted by receiver, CSR offsets), with the edges' residual update
 * All pointers are device pointers; calls are asynchronous on `stream` (a hipStream_t); nothing is allocated inside. */
#ifndef SKYRIM_GRAPHCAST_H
#define SKYRIM_GRAPHCAST_H

#ifdef __cplusplus
extern "C" {
#endif

#define SKGC_ABI_VERSION 5
#define SKGC_E_ARG (-1)
#define SKGC_E_HIP (-2)

/* out[m][n] = act( sum_k A(m, k) W[n][k] + bias[n] ),  m < M, n < N, out row-major with leading dimension ldo.
 * A(m, :) = concat_s src[s][ idx[s] ? idx[s][m] : m ][0 .. width[s])   (fp32 rows with leading dimension ld[s]; widths are
 * multiples of 8 except the last; K = sum of widths); optional per-k affine A * kscale[k] + kshift[k] before the fp16 split.
 * W: fp16 hi/lo planes [N][ldw] prepared by sksfno_prepare_weight. */
typedef struct skgc_gather_gemm_desc {
    const float* src[3];
    const int* idx[3];
    long long ld[3];
    int width[3];
    int n_src;
    const float* kscale;
    const float* kshift;
    const void* w;
    long long w_plane;
    int ldw;
    const float* bias;
    float* out;
    long long ldo;
    int M, N;
    int act;   /* 0 = none, 2 = swish */
} skgc_gather_gemm_desc;

int skgc_abi_version(void);
int skgc_gather_gemm(const skgc_gather_gemm_desc* desc, void* stream);

/* out[r][:] = (res ? res[r][:] : 0) + LayerNorm(x[r][:]) * gamma + beta  over N columns (eps 1e-5); out may alias res or x */
int skgc_layer_norm(const float* x, const float* gamma, const float* beta, const float* res, float* out, long long rows, int N, void* stream);

/* out[v][:] = sum of e[j][:] for offsets[v] <= j < offsets[v + 1]   (edges sorted by receiver; nodes without edges get zeros);
 * acc (nullable): acc[j][:] += e[j][:] for every edge row on the way (the residual update of the edge latents) */
int skgc_segment_sum(const float* e, const int* offsets, float* out, float* acc, int n_nodes, int N, void* stream);

/* Second Linear of an MLP fused with its LayerNorm, latent width 512 only:
 *   out[r][:] = (res ? res[r][:] : 0) + LayerNorm( a[r][0..K) W^T + bias ) * gamma + beta,   r < rows, 512 columns, eps 1e-5
 * a: fp32 rows with leading dimension lda; W: fp16 hi/lo planes [512][ldw] from skgc_prepare_weight_perm8; out may alias res. */
int skgc_prepare_weight_perm8(const float* src, int N, int K, void* dst, long long plane, int ldw, void* stream);
int skgc_linear_layer_norm(const float* a, long long lda, int K, const void* w, long long w_plane, int ldw, const float* bias, const float* gamma,
                           const float* beta, const float* res, float* out, long long rows, void* stream);

/* Edge MLP with the first Linear taken apart by distributivity (512 columns only):
 *   out[r][:] = (res ? res[r][:] : 0) + LayerNorm( act( sum_s src[s][ idx[s] ? idx[s][r] : r ][0..K) ) W^T + bias ) * gamma + beta
 * i.e. the caller has computed the per-edge term e W_e^T (+ fc1 bias) and the per-NODE terms v_s W_s^T, v_r W_r^T; this kernel
 * gathers and adds the three rows, applies the activation and runs the second Linear + LayerNorm (+ residual).  Replaces
 * skgc_gather_gemm over concat(e, v_s[send], v_r[recv]) followed by skgc_linear_layer_norm: same result up to fp32 rounding.
 * src rows are fp32 with leading dimension ld[s] >= K, 16-byte aligned; K % 8 == 0; W as for skgc_linear_layer_norm. */
typedef struct skgc_sum_desc {
    const float* src[3];
    const int* idx[3];
    long long ld[3];
    int n_src;
    int K;
    int act;                 /* 0 = none, 2 = swish */
    const void* w;
    long long w_plane;
    int ldw;
    const float* bias;       /* second Linear's bias, [512] or NULL */
    const float* gamma;
    const float* beta;
    const float* res;        /* NULL or [rows][512]; out may alias res */
    float* out;              /* [rows][512] */
    long long rows;
    /* ABI v3.  group == 3: `rows` counts GROUPS of three input rows and out[g] = sum over the group's members of the LayerNorm
     * output (the receiver sum of the mesh->grid edges: three edges into every grid node).  The input has 48 * ceil(rows / 16)
     * virtual rows -- row 48 t + 16 a + l is member a of group 16 t + l -- and every source is addressed through its index
     * array in that order (entries of groups >= rows: any valid row); res must be NULL.  0 / 1: one output row per input row. */
    int group;
} skgc_sum_desc;
int skgc_sum_linear_layer_norm(const skgc_sum_desc* desc, void* stream);

/* ---- ABI v4: an interaction-network update as ONE kernel (csrc/graphcast_fused.hip; layouts: skyrim_amd/graphcast/fused.py) ----------
 *
 * Edge update + receiver sum, latent 512, on PACKED rows (multiples of 128; a receiver's run of rows never crosses a 128-row tile unless
 * it is longer than a tile; padding rows have recv < 0 and idx < 0):
 *     pre  = (has_fc1 ? e_in W_e^T : e_in) + sum_s term[s][ idx[s][row] ]            first Linear by distributivity (b1 folded into a term)
 *     y    = LayerNorm( swish(pre) W2^T + b2 ) * gamma + beta
 *     e_out[row] = e_in[row] + y   (has_fc1 only, nullable)          agg[ recv[row] ] = sum of y over the receiver's run of rows
 * e_in / e_out: ONE fp16 plane in the blocked layout [rows/16][16][16][32] (may alias); with has_fc1 == 0, e_in is the prepared
 * first-Linear term in "pos" column order.  term[s]: fp32 rows, leading dimension ld[s], "pos" column order.  w1f / w2f: fp16 planes in
 * MFMA fragment order (fused.py: prep_w1_fragments / prep_w2_fragments); w2f always hi/lo (two MFMA terms W_hi h + W_lo h: activations
 * are rounded to fp16, the second Linear's weights are not), w1f with w1_planes planes.  A tile whose first run continues the previous tile's last run writes that run's sum to
 * heads[tile] instead of agg; skgc_segment_fixup adds those pieces (agg[nodes[i]] += heads[tiles[k]], first[i] <= k < first[i + 1],
 * in that order).  Every receiver with at least one row is written exactly once per call; deterministic (no atomics). */
typedef struct skgc_edge_desc {
    const void* e_in;
    void* e_out;
    const float* term[2];
    const int* idx[2];
    long long ld[2];
    int n_term;              /* 0..2 */
    const int* recv;
    const void* w1f;
    const void* w2f;
    const float* b2;
    const float* gamma;
    const float* beta;
    float* agg;
    float* heads;            /* [rows / 128][512].  NULL only if NO tile's first receiver equals the previous tile's last (recv[128 t] != recv[128 t - 1]
                              * for every t): the piece of a run that continues into a tile is then dropped, not written (never a wild store) */
    long long rows;
    int has_fc1;
    int w1_planes;           /* has_fc1: planes of w1f -- 2 = fp16 hi/lo (two MFMA terms), 1 = W_e rounded to fp16 (one term) */
    long long* probe;        /* NULL, or [rows / 128][8]: shader clocks of the tile's first wave at the phase boundaries (measurement only) */
} skgc_edge_desc;
int skgc_edge_update(const skgc_edge_desc* desc, void* stream);
int skgc_segment_fixup(float* agg, const float* heads, const int* nodes, const int* first, const int* tiles, int n_nodes, void* stream);

/* Node update on fp32 rows, latent 512, three MFMA terms (fp32 operands split into fp16 hi/lo on the fly):
 *     out[r] = (res ? res[r] : 0) + LayerNorm( swish( concat_s src[s][r] W1^T + b1 ) W2^T + b2 ) * gamma + beta,   r < rows
 * src[s]: fp32 rows of 512 columns (n_src = 1 or 2: W1 is [512][512 n_src]); out may alias res or a source.
 * w1f (ABI 5: the layout changed from chunk order to K-OUTER with round 5's kernel, so a library of either generation refuses the other's
 * blobs through the version): per source s, 1 KiB fragments in the order [ks = 0..15][n = 0..31][plane hi, lo] -- fragment (ks, n) holds,
 * lane-linear (lane l, element e), W1[32 (n >> 1) + hidden unit of (n & 1, l & 15)][512 s + 32 ks + 8 (l >> 4) + e]: ONE k-step of all 512
 * hidden units is one 64 KiB LDS stage (skyrim_amd/graphcast/fused.py: prep_w1_fragments_kouter is the reference packer).
 * w2f: [j = 0..15][c = 0..31][plane] fragments of the second Linear, perm8 rows (fused.py: prep_w2_fragments). */
typedef struct skgc_node_desc {
    const float* src[2];
    long long ld[2];
    int n_src;
    const void* w1f;
    const void* w2f;
    const float* b1;
    const float* b2;
    const float* gamma;
    const float* beta;
    const float* res;
    long long ld_res;
    float* out;
    long long ld_out;
    long long rows;
} skgc_node_desc;
int skgc_node_mlp(const skgc_node_desc* desc, void* stream);

#ifdef __cplusplus
}
#endif
#endif
