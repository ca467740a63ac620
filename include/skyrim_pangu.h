/*
 * skyrim_pangu.h -- C ABI of the MI355X (gfx950) Pangu-Weather 6-h step engine.
 * Drop-in boundary.  The reference has no C ABI: its seam is the duck-typed earth2mip ``TimeLoop`` returned by ``PanguModel.build_model()``
 * (/root/reference/skyrim/core/models/pangu.py:45-46) and driven by ``run_basic_inference`` (/root/reference/skyrim/core/models/utils.py:34:
 * ``for k, (time, output, _) in enumerate(model(time, x))``).  One iteration of that generator is one skpangu_step() here; the Python object
 * that re-creates the TimeLoop protocol on this library is skyrim_amd/pangu/timeloop.py (the binding a maintainer would add).
 * Conventions: every *_dev pointer is DEVICE memory owned by the caller; the library never allocates, frees or synchronises; ``stream`` is a
 * hipStream_t passed as void*; all work is stream-ordered and re-entrant across contexts.  State tensors are float32 [69][n_lat][n_lon] in the
 * channel order of /root/reference/skyrim/core/models/pangu.py:6-13 (z,q,t,u,v x 1000..50 hPa, then msl,u10m,v10m,t2m), lat 90..-90,
 * lon 0..360, physical units (normalisation constants are model parameters).  Returns 0, a positive hipError_t, or a negative SKPANGU_E_*.
 */
#ifndef SKYRIM_PANGU_H
#define SKYRIM_PANGU_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SKPANGU_ABI_VERSION 6

/* precision modes: how each matrix product is formed on the MFMA pipe (values 2 and 5 of ABI v4, the fp16-hidden variants, are gone) */
#define SKPANGU_PREC_BF16X3 0 /* bf16 hi/lo split, 3 MFMA terms, fp32 range (wide-range mode) */
/* (value 1, a single fp16 term everywhere, was a speed probe outside the 1e-3 bar and slower than the default plan: removed in ABI v6) */
#define SKPANGU_PREC_F16X3  3 /* fp16 hi/lo split, 3 MFMA terms (22-bit operands; activations must stay < 65504) */
#define SKPANGU_PREC_F16X3_Q 4 /* f16x3 with the QKV linear reading only the hi plane of the stream; with a term_plan: the host's modes */

#define SKPANGU_E_ARG        (-1) /* bad argument / unsupported geometry */
#define SKPANGU_E_SIZE       (-2) /* a caller buffer is too small */
#define SKPANGU_E_STATE      (-3) /* skpangu_prepare() has not run */
#define SKPANGU_E_NOTFOUND   (-4)

typedef struct skpangu_ctx skpangu_ctx;

/* Conventions the public Pangu pseudocode leaves open are configuration here and in the CPU oracle (oracle/pangu_oracle.py: Conventions);
 * 0 selects the default.  They are applied once, in skpangu_prepare: no kernel depends on them. */
#define SKPANGU_PAD_CENTRE 0 /* zero padding split front = total / 2, back = rest (default) */
#define SKPANGU_PAD_BACK   1 /* all zero padding behind the data */

typedef struct skpangu_config {
    int n_lat;     /* 721; any n_lat >= 8 */
    int n_lon;     /* 1440; must be a multiple of 96 */
    int precision; /* SKPANGU_PREC_* */
    int roll_sign; /* shifted-window blocks: -1 (default, also 0) roll by -(1,3,6) first (Swin); +1 as the pseudocode's call is written */
    int pad_mode;  /* SKPANGU_PAD_* for every zero padding on the path (input latitude, window latitude, 2x2 merge) */
    float mask_value; /* additive shifted-window mask; 0 = default -100 (Swin) */
    int mlp_mode;  /* 0 (default): the row-tile kernels (proj + LayerNorm + MLP + LayerNorm of a block as ONE kernel, row-tile QKV);
                      1: every linear as a tiled LDS-DMA GEMM (the round-1 path; the calibration taps run on it) */
    int term_plan; /* per-layer MFMA term plan of the fp16-plane modes with mlp_mode 0 (0: three terms everywhere).  Bit l (l = 0..3): the blocks
                      of layer l + 1 run proj / fc1 / fc2 with the weights as ONE fp16 plane, two terms A_hi W + A_lo W.  Bit 4 + l (F16X3_Q):
                      the layer's QKV with ONE term.  Bit 8 + l (needs bit l): proj / fc1 / fc2 of the layer with ONE term -- the activation
                      operands enter the GEMMs as their fp16 hi plane too; the residual path keeps hi/lo pairs.  Host modes
                      (skyrim_amd/pangu/engine.py): "f16x1m" (default) = 0x66F, "f16x2m" = 0x6F, "f16x2c" = 0x66; numbers: DESIGN.md 3. */
    int surface_last;    /* 0 (default): the surface slab is token level 0; 1: it is the LAST level (PatchEmbedding's concatenate as written) */
    int qkv_order;       /* packing of the qkv Linear's 3C output rows in the MASTER weights: 0 (default) (3, heads, head_dim); 1 (heads, 3, head_dim) */
    int bias_transposed; /* 0 (default): bias gathered as [query][key] from position_index; 1: [key][query] */
} skpangu_config;

typedef struct skpangu_sizes {
    long long master_floats;   /* elements of the fp32 master parameter blob (skpangu_param_info layout) */
    size_t prepared_bytes;     /* kernel-ready parameter arena (16-bit weight planes, expanded bias, tables) */
    size_t workspace_bytes;    /* activations of one step (residual streams, Q/K/V, MLP hidden) */
    long long state_floats;    /* 69 * n_lat * n_lon */
    int n_params;              /* entries of the master parameter table */
} skpangu_sizes;

int skpangu_abi_version(void);
const char* skpangu_error_string(int code);
/* Buffer sizes for a configuration; no GPU needed. */
int skpangu_query_sizes(const skpangu_config* cfg, skpangu_sizes* out);
/* Master parameter table (the fp32 blob handed to skpangu_prepare): entry i is tensor ``name`` with ``ndim`` dims ``shape`` at element offset
 * ``offset``; names mirror a PyTorch state dict of the public pseudocode ("layer2.block3.attn.qkv.weight").  No GPU needed. */
int skpangu_param_info(const skpangu_config* cfg, int index, char* name, size_t name_cap, long long* offset, int* ndim, long long shape[6]);
/* Create a context over caller-owned device buffers (sizes from skpangu_query_sizes). */
int skpangu_create(const skpangu_config* cfg, void* prepared_dev, size_t prepared_bytes, void* workspace_dev, size_t workspace_bytes, skpangu_ctx** out);
void skpangu_destroy(skpangu_ctx* ctx);
/* One-time conversion of the fp32 master blob into the prepared arena (weight planes in fragment order, earth-specific bias per window type with
 * the shifted-window mask folded in, window gather tables).  Replaces the ONNX-session construction of earth2mip.networks.pangu.load.  One-plane
 * weights are rounded to nearest; master weights already on the fp16 grid pass through (the host's compensated rounding hands such weights in). */
int skpangu_prepare(skpangu_ctx* ctx, const float* master_dev, void* stream);
/* Calibration of a term plan (a no-op without one), after skpangu_prepare with the same master blob: one step on ``state_in_dev`` through the
 * three-term kernels, the column means of the operand of every Linear the plan runs short, and (W - fp16(W)) x mean added to that Linear's
 * prepared bias.  NULL state: back to the master biases.  No counterpart in the reference: it belongs to the operand format. */
int skpangu_calibrate(skpangu_ctx* ctx, const float* master_dev, const float* state_in_dev, void* stream);
/* One 6-h forecast step: state_out = Pangu6(state_in); in place allowed.  Replaces one iteration of the reference's TimeLoop generator
 * (/root/reference/skyrim/core/models/utils.py:34). */
int skpangu_step(skpangu_ctx* ctx, const float* state_in_dev, float* state_out_dev, void* stream);
/* Stage-level entry points (same kernels as skpangu_step; parity tests and calibration taps).  Token tensors are float32 row-major
 * [tokens][channels]: x1 / x4 / skip [8 H1 W1][192], x2 [8 H2 W2][384]. */
int skpangu_patch_embed(skpangu_ctx* ctx, const float* state_in_dev, float* x1_dev, void* stream);
int skpangu_block(skpangu_ctx* ctx, int layer /*1..4*/, int block, float* x_inout_dev, void* stream);
int skpangu_downsample(skpangu_ctx* ctx, const float* x1_dev, float* x2_dev, void* stream);
int skpangu_upsample(skpangu_ctx* ctx, const float* x2_dev, float* x4_dev, void* stream);
int skpangu_patch_recover(skpangu_ctx* ctx, const float* skip_dev, const float* x4_dev, float* state_out_dev, void* stream);
/* Per-stage timing with HIP events recorded on the launch stream between the launches of skpangu_step (bench.py's roofline leg).  flops / bytes
 * are the ALGORITHMIC work of one launch (2 M N K of the model's GEMM; minimum HBM traffic at the mode's storage types). */
typedef struct skpangu_stage_stat {
    char name[24];
    int launches;      /* launches timed since skpangu_profile(ctx, 1) / the last read */
    double total_ms;   /* sum of their durations */
    double flops;      /* per launch */
    double bytes;      /* per launch */
} skpangu_stage_stat;
int skpangu_profile(skpangu_ctx* ctx, int enable);
int skpangu_profile_read(skpangu_ctx* ctx, skpangu_stage_stat* out, int cap, int* n);
/* Debug view of an internal buffer ("q","k","vt","ao","hid","u","x1","x2","x4","widx<res><roll>","bias_exp<blk>"); owned by the context's arenas. */
int skpangu_debug_buffer(skpangu_ctx* ctx, const char* name, void** ptr_dev, size_t* bytes);

#ifdef __cplusplus
}
#endif
#endif /* SKYRIM_PANGU_H */
