// C ABI (include/skyrim_pangu.h) over the stage launchers: geometry, master-parameter table,
// arena planning, prepare, and the fixed launch sequence of one Pangu 6-h step.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/skyrim_pangu.h"
#include "launchers.h"

using namespace skp;

namespace {

constexpr int kDepths[4] = {2, 6, 6, 2};
constexpr int kBiasRows = 3312;

int pad_to(int n, int mult, int* front, int pad_mode) {
    const int padded = (n + mult - 1) / mult * mult;
    if (front) *front = pad_mode == SKPANGU_PAD_CENTRE ? (padded - n) / 2 : 0;
    return padded;
}

bool make_geom(const skpangu_config& c, Geom& g) {
    if (c.n_lat < 8 || c.n_lon <= 0 || c.n_lon % 96 != 0) return false;
    if (c.roll_sign < -1 || c.roll_sign > 1 || (c.pad_mode != SKPANGU_PAD_CENTRE && c.pad_mode != SKPANGU_PAD_BACK)) return false;
    if (c.mask_value > 0.f || c.mask_value < -60000.f) return false;      // the mask lives in the fp16 bias tiles
    if (c.mlp_mode != 0 && c.mlp_mode != 1) return false;
    if (c.term_plan < 0 || c.term_plan > 0xFFF) return false;
    if (((c.term_plan >> 8) & 0xF) & ~(c.term_plan & 0xF)) return false;      // a one-term layer is a two-term layer that also drops the activations' lo planes
    if ((c.surface_last | c.qkv_order | c.bias_transposed) & ~1) return false;
    g.surface_last = c.surface_last; g.qkv_order = c.qkv_order; g.bias_transposed = c.bias_transposed;
    g.roll_sign = c.roll_sign > 0 ? 1 : -1;
    g.mask_value = c.mask_value == 0.f ? -100.f : c.mask_value;
    g.n_lat = c.n_lat; g.n_lon = c.n_lon; g.n_levels = 13; g.n_channels = 69; g.surf0 = 65;
    const int latp = pad_to(c.n_lat, 4, &g.lat_top, c.pad_mode);
    g.Z = 8; g.H1 = latp / 4; g.W1 = c.n_lon / 4;
    g.H2 = pad_to(g.H1, 2, nullptr, c.pad_mode) / 2; g.W2 = g.W1 / 2;       // 2x2 merge: an odd H1 pads ONE row, behind the data in both modes
    const int H[2] = {g.H1, g.H2}, W[2] = {g.W1, g.W2};
    for (int r = 0; r < 2; ++r) {
        g.Hp[r] = pad_to(H[r], 6, &g.top[r], c.pad_mode);
        g.nH[r] = g.Hp[r] / 6; g.nW[r] = W[r] / 12;
        g.types[r] = (g.Z / 2) * g.nH[r];
        g.nwin[r] = g.types[r] * g.nW[r];
        g.ntok[r] = g.Z * H[r] * W[r];
        g.mwin[r] = g.nwin[r] * WIN_TOKENS;
    }
    return true;
}

struct Param { std::string name; std::vector<long long> shape; long long offset; long long numel; };

std::vector<Param> build_params(const Geom& g, long long* total) {
    std::vector<Param> v;
    long long off = 0;
    auto add = [&](const std::string& n, std::vector<long long> s) {
        long long ne = 1;
        for (long long d : s) ne *= d;
        v.push_back(Param{n, s, off, ne});
        off += (ne + 63) / 64 * 64;
    };
    add("norm.mean", {69}); add("norm.std", {69});
    add("const_masks", {3, g.n_lat, g.n_lon});
    add("embed.conv.weight", {192, 5, 2, 4, 4}); add("embed.conv.bias", {192});
    add("embed.conv_surface.weight", {192, 7, 4, 4}); add("embed.conv_surface.bias", {192});
    for (int layer = 0; layer < 4; ++layer) {
        const int c = layer_dim(layer), heads = layer_heads(layer), types = g.types[layer_res(layer)];
        for (int i = 0; i < kDepths[layer]; ++i) {
            const std::string p = "layer" + std::to_string(layer + 1) + ".block" + std::to_string(i) + ".";
            add(p + "attn.bias_table", {kBiasRows, types, heads});
            add(p + "attn.qkv.weight", {3 * c, c}); add(p + "attn.qkv.bias", {3 * c});
            add(p + "attn.proj.weight", {c, c}); add(p + "attn.proj.bias", {c});
            add(p + "norm1.weight", {c}); add(p + "norm1.bias", {c});
            add(p + "mlp.fc1.weight", {4 * c, c}); add(p + "mlp.fc1.bias", {4 * c});
            add(p + "mlp.fc2.weight", {c, 4 * c}); add(p + "mlp.fc2.bias", {c});
            add(p + "norm2.weight", {c}); add(p + "norm2.bias", {c});
        }
        if (layer == 0) { add("down.norm.weight", {768}); add("down.norm.bias", {768}); add("down.linear.weight", {384, 768}); }
        if (layer == 2) { add("up.linear1.weight", {768, 384}); add("up.norm.weight", {192}); add("up.norm.bias", {192}); add("up.linear2.weight", {192, 192}); }
    }
    add("recover.conv.weight", {384, 5, 2, 4, 4}); add("recover.conv.bias", {5});
    add("recover.conv_surface.weight", {384, 4, 4, 4}); add("recover.conv_surface.bias", {4});
    if (total) *total = off;
    return v;
}

struct Arena {
    char* base;
    size_t off;
    template <class U> U* take(size_t n) {
        off = (off + 255) / 256 * 256;
        U* p = base ? reinterpret_cast<U*>(base + off) : nullptr;
        off += n * sizeof(U);
        return p;
    }
};

struct IEngine {
    virtual ~IEngine() {}
    virtual hipError_t prepare(const float* master, hipStream_t s) = 0;
    virtual hipError_t step(const float* in, float* out, hipStream_t s) = 0;
    virtual hipError_t calibrate(const float* master, const float* in, hipStream_t s) = 0;
    virtual hipError_t embed(const float* in, float* x1, hipStream_t s) = 0;
    virtual hipError_t block(int layer, int blk, float* x, hipStream_t s) = 0;
    virtual hipError_t down(const float* x1, float* x2, hipStream_t s) = 0;
    virtual hipError_t up(const float* x2, float* x4, hipStream_t s) = 0;
    virtual hipError_t recover(const float* skip, const float* x4, float* out, hipStream_t s) = 0;
    virtual bool debug(const std::string& name, void** p, size_t* bytes) = 0;
    virtual void profile(bool on) = 0;
    virtual hipError_t profile_read(skpangu_stage_stat* out, int cap, int* n) = 0;
    virtual size_t prepared_bytes() const = 0;
    virtual size_t workspace_bytes() const = 0;
    virtual void bind(char* prepared, char* workspace) = 0;
};

template <class P>
struct Engine : IEngine {
    typedef typename P::T T;
    static constexpr int NW = P::NW;
    static constexpr int NPL = (P::NA > P::NW ? P::NA : P::NW);

    Geom g;
    ModelW<T> w;
    Work<P> wk;
    std::vector<Param> params;
    size_t prep_bytes = 0, ws_bytes = 0;
    size_t bias_exp_elems[16];
    size_t q_elems = 0, ao_elems = 0, hid_elems = 0;
    bool fused_mlp = false;   // one-kernel MLP (fused_mlp.hip): 3-term modes with the hidden as hi/lo pair
    bool rt_proj = false, rt_qkv = false;   // row-tile proj / QKV kernels (rowtile.hip)
    bool fused_block = false;               // proj + LayerNorm + residual + MLP as one kernel (fused_block.hip)
    bool attn2 = true;                      // attention with the bias gathered from the compact table in LDS (SKP_ATTN_V1=1: the expanded table)
    bool fused_qa = true;                   // QKV + attention as ONE kernel, q / k / v in registers (attention.hip; SKP_SPLIT_ATTN=1: the two launches)
    int plan2 = 0;                          // bit l: layer l + 1 runs proj / fc1 / fc2 with TWO terms (weights as one fp16 plane, fused_block2.hip)
    bool two_term(int layer) const { return (plan2 >> layer) & 1; }
    bool qkv_one(int layer) const { return rt_qkv && ((plan2 >> (4 + layer)) & 1); }   // QKV with ONE term (stream hi plane x weight hi plane)
    bool block_one(int layer) const { return (plan2 >> (8 + layer)) & 1; }             // proj / fc1 / fc2 with ONE term (activation hi plane x weight hi plane)
    // the fused QKV + attention kernel takes the row-tile QKV weights of a block: one plane at either width, hi / lo planes at C = 192 only (LDS)
    bool qa_fused(const BlockW<T>& bw, int res) const { return fused_qa && rt_qkv && attn2 && bw.bias_cmp && (bw.qkvh || (bw.qkvf && res == 0)); }
    T* zrow = nullptr;
    float *qkv_w_tmp = nullptr, *qkv_b_tmp = nullptr;
    float* cal_sum = nullptr;               // column sums of one GEMM operand (calibrate)

    // ---- per-stage timing with HIP events on the launch stream (bench.py roofline leg) ---- //
    enum Cat { C_EMBED, C_QKV0, C_ATTN0, C_PROJ0, C_FC1_0, C_FC2_0, C_QKV1, C_ATTN1, C_PROJ1, C_FC1_1, C_FC2_1, C_DOWN, C_UP, C_RECOVER, C_COUNT };
    bool prof_on = false;
    std::vector<hipEvent_t> prof_ev;
    std::vector<int> prof_cat;
    size_t prof_used = 0;
    void mark(int cat, hipStream_t s) {
        if (!prof_on) return;
        if (prof_used == prof_ev.size()) {
            hipEvent_t e;
            if (hipEventCreate(&e) != hipSuccess) return;
            prof_ev.push_back(e);
            prof_cat.push_back(-1);
        }
        prof_cat[prof_used] = cat;
        (void)hipEventRecord(prof_ev[prof_used], s);
        ++prof_used;
    }
    void profile(bool on) override { prof_on = on; prof_used = 0; }
    ~Engine() override { for (hipEvent_t e : prof_ev) (void)hipEventDestroy(e); }
    hipError_t profile_read(skpangu_stage_stat* out, int cap, int* n) override {
        static const char* names_split[C_COUNT] = {"embed", "qkv_r0", "attn_r0", "proj_r0", "fc1_r0", "fc2_r0", "qkv_r1", "attn_r1",
                                                   "proj_r1", "fc1_r1", "fc2_r1", "downsample", "upsample", "recover"};
        static const char* names_fused[C_COUNT] = {"embed", "qkv_r0", "attn_r0", "proj_r0", "mlp_r0", "fc2_r0", "qkv_r1", "attn_r1",
                                                   "proj_r1", "mlp_r1", "fc2_r1", "downsample", "upsample", "recover"};
        static const char* names_block[C_COUNT] = {"embed", "qkv_r0", "attn_r0", "proj_r0", "proj_mlp_r0", "fc2_r0", "qkv_r1", "attn_r1",
                                                   "proj_r1", "proj_mlp_r1", "fc2_r1", "downsample", "upsample", "recover"};
        const char* const* names = fused_block ? names_block : (fused_mlp ? names_fused : names_split);
        double ms[C_COUNT] = {0}; int cnt[C_COUNT] = {0};
        if (prof_used > 0) {
            hipError_t e = hipEventSynchronize(prof_ev[prof_used - 1]);
            if (e != hipSuccess) return e;
        }
        for (size_t i = 0; i + 1 < prof_used; ++i) {
            if (prof_cat[i] < 0) continue;
            float t = 0.f;
            hipError_t e = hipEventElapsedTime(&t, prof_ev[i], prof_ev[i + 1]);
            if (e != hipSuccess) return e;
            ms[prof_cat[i]] += t; cnt[prof_cat[i]]++;
        }
        const double sa = 2.0 * NPL, hw = (double)g.H1 * g.W1, NA_ = P::NA;
        double fl[C_COUNT], by[C_COUNT];
        for (int r = 0; r < 2; ++r) {
            const double C = r == 0 ? 192 : 384, mw = g.mwin[r], nt = g.ntok[r], heads = C / 32, wb = 2.0 * NW;
            const int o = r == 0 ? 0 : 5;
            fl[C_QKV0 + o] = 2 * mw * C * 3 * C;      by[C_QKV0 + o] = nt * C * 2 * NA_ + 3 * mw * C * 2 + 3 * C * C * wb;
            fl[C_ATTN0 + o] = g.nwin[r] * heads * 4.0 * 144 * 144 * 32;
            by[C_ATTN0 + o] = 3 * mw * C * 2 + mw * C * sa + (double)g.types[r] * heads * (attn2 ? 3456 : 81 * 256) * 2;
            fl[C_PROJ0 + o] = 2 * mw * C * C;          by[C_PROJ0 + o] = mw * C * sa + 2 * nt * C * 4 + C * C * wb;   // stream: 4 B/elem read + 4 B/elem written
            fl[C_FC1_0 + o] = 2 * nt * C * 4 * C;      by[C_FC1_0 + o] = nt * C * 2 * NA_ + nt * 4 * C * sa + 4 * C * C * wb;
            fl[C_FC2_0 + o] = 2 * nt * C * 4 * C;      by[C_FC2_0 + o] = nt * 4 * C * sa + 2 * nt * C * 4 + 4 * C * C * wb;
            if (fused_qa && rt_qkv && attn2 && (r == 0 || (plan2 >> (4 + 1)) & 1)) {   // QKV inside the attention launch: its FLOPs, the stream's hi plane in, the output out
                fl[C_ATTN0 + o] += fl[C_QKV0 + o];
                by[C_ATTN0 + o] = nt * C * 2 + mw * C * sa + 3 * C * C * wb + (double)g.types[r] * heads * 3456 * 2;
            }
            if (fused_mlp) {   // one kernel: both GEMMs, stream read once + written once, both weight matrices
                fl[C_FC1_0 + o] = 4 * nt * C * 4 * C;  by[C_FC1_0 + o] = 2 * nt * C * 4 + 8 * C * C * wb;
            }
            if (fused_block) { // ... and the projection in front of them: + its FLOPs, + the attention rows read, + its weights
                fl[C_FC1_0 + o] += 2 * mw * C * C;     by[C_FC1_0 + o] += nt * C * sa + C * C * wb;
            }
        }
        fl[C_EMBED] = 2 * hw * 112 * 192 + 2 * 7 * hw * 160 * 192; by[C_EMBED] = (69.0 + 3) * g.n_lat * g.n_lon * 4 + g.ntok[0] * 192.0 * 4;
        fl[C_DOWN] = 2.0 * g.ntok[1] * 768 * 384;                   by[C_DOWN] = g.ntok[0] * 192.0 * 4 + g.ntok[1] * 384.0 * 4;
        fl[C_UP] = 2.0 * g.ntok[1] * 384 * 768 + 2.0 * g.ntok[0] * 192 * 192;
        by[C_UP] = g.ntok[1] * 384.0 * 4 + 2 * g.ntok[0] * 192.0 * sa + g.ntok[0] * 192.0 * 4;
        fl[C_RECOVER] = 2 * hw * 384 * 64 + 2 * 7 * hw * 384 * 160; by[C_RECOVER] = 2 * g.ntok[0] * 192.0 * 4 + 69.0 * g.n_lat * g.n_lon * 4;
        int k = 0;
        for (int c = 0; c < C_COUNT && k < cap; ++c, ++k) {
            std::memset(&out[k], 0, sizeof(out[k]));
            std::strncpy(out[k].name, names[c], sizeof(out[k].name) - 1);
            out[k].launches = cnt[c]; out[k].total_ms = ms[c]; out[k].flops = fl[c]; out[k].bytes = by[c];
        }
        *n = k;
        prof_used = 0;
        return hipSuccess;
    }

    LinW<T> take_lin(Arena& a, int N, int ldd) {
        LinW<T> l;
        const long long plane = (long long)N * ldd;
        l.w = a.take<T>((size_t)plane * NW);
        l.plane = plane; l.ldw = ldd;
        return l;
    }

    void plan_prepared(char* base) {
        Arena a{base, 0};
        w.mean = a.take<float>(69); w.std = a.take<float>(69); w.istd = a.take<float>(69);
        w.masks = a.take<float>((size_t)3 * g.n_lat * g.n_lon);
        w.embed_u = take_lin(a, 192, 160); w.embed_s = take_lin(a, 192, 128);
        w.embed_u_b = a.take<float>(192); w.embed_s_b = a.take<float>(192);
        int b = 0;
        for (int layer = 0; layer < 4; ++layer) {
            const int c = layer_dim(layer), heads = layer_heads(layer), res = layer_res(layer);
            for (int i = 0; i < kDepths[layer]; ++i, ++b) {
                BlockW<T>& bw = w.blk[b];
                bw.qkv = take_lin(a, 3 * c, c); bw.proj = take_lin(a, c, c);
                bw.fc1 = take_lin(a, 4 * c, c); bw.fc2 = take_lin(a, c, 4 * c);
                const bool t2 = two_term(layer);
                bw.w1f = fused_mlp && !t2 ? a.take<T>((size_t)8 * c * c) : nullptr;      // 4c x c elements, hi + lo
                bw.w2f = fused_mlp && !t2 ? a.take<T>((size_t)8 * c * c) : nullptr;
                bw.projf = (rt_proj || fused_block) && !t2 ? a.take<T>((size_t)2 * c * c) : nullptr;
                bw.projh = t2 ? a.take<T>((size_t)c * c) : nullptr;                       // hi plane only
                bw.w1h = t2 ? a.take<T>((size_t)4 * c * c) : nullptr;
                bw.w2h = t2 ? a.take<T>((size_t)4 * c * c) : nullptr;
                bw.qkvf = rt_qkv && !qkv_one(layer) ? a.take<T>((size_t)6 * c * c) : nullptr;
                bw.qkvh = qkv_one(layer) ? a.take<T>((size_t)3 * c * c) : nullptr;
                bw.qkv_b = a.take<float>(3 * c); bw.proj_b = a.take<float>(c);
                bw.fc1_b = a.take<float>(4 * c); bw.fc2_b = a.take<float>(c);
                bw.n1_g = a.take<float>(c); bw.n1_b = a.take<float>(c);
                bw.n2_g = a.take<float>(c); bw.n2_b = a.take<float>(c);
                bias_exp_elems[b] = attn2 ? (size_t)g.types[res] * heads * 3456 : (size_t)g.types[res] * heads * 81 * 256;
                bw.bias_exp = attn2 ? nullptr : a.take<f16>(bias_exp_elems[b]);
                bw.bias_cmp = attn2 ? a.take<f16>(bias_exp_elems[b]) : nullptr;
            }
        }
        w.down_g = a.take<float>(768); w.down_b = a.take<float>(768);
        w.down = take_lin(a, 384, 768);
        w.up1 = take_lin(a, 768, 384); w.up2 = take_lin(a, 192, 192);
        w.up_g = a.take<float>(192); w.up_b = a.take<float>(192);
        w.rec_u = take_lin(a, 160, 384); w.rec_s = take_lin(a, 64, 384);
        w.rec_u_b = a.take<float>(5); w.rec_s_b = a.take<float>(4);
        for (int r = 0; r < 2; ++r)
            for (int roll = 0; roll < 2; ++roll) { w.widx[r][roll] = a.take<int>(g.mwin[r]); w.winv[r][roll] = a.take<int>(g.ntok[r]); }
        zrow = a.take<T>(4096);
        qkv_w_tmp = a.take<float>((size_t)3 * 384 * 384); qkv_b_tmp = a.take<float>(3 * 384);      // canonical-order copy of a qkv Linear (qkv_order = 1)
        cal_sum = a.take<float>(4 * 384);
        prep_bytes = (a.off + 255) / 256 * 256;
    }

    void plan_workspace(char* base) {
        Arena a{base, 0};
        const size_t n0 = (size_t)g.ntok[0] * 192, n1 = (size_t)g.ntok[1] * 384;
        wk.xs_plane[0] = (long long)n0; wk.xs_plane[1] = (long long)n1;
        wk.X1s = a.take<T>(n0 * 2); wk.X2s = a.take<T>(n1 * 2); wk.X4s = a.take<T>(n0 * 2);     // always hi + lo
        const size_t m0 = (size_t)g.mwin[0] * 192, m1 = (size_t)g.mwin[1] * 384;
        q_elems = m0 > m1 ? m0 : m1;
        wk.qkv_plane = (long long)q_elems;
        wk.q = a.take<f16>(q_elems); wk.k = a.take<f16>(q_elems); wk.vt = a.take<f16>(q_elems);
        ao_elems = q_elems;
        wk.ao_plane = (long long)ao_elems;
        wk.ao = a.take<T>(ao_elems * NPL);
        const size_t h0 = (size_t)g.ntok[0] * 768, h1 = (size_t)g.ntok[1] * 1536;
        hid_elems = h0 > h1 ? h0 : h1;
        wk.hid_plane = (long long)hid_elems;
        wk.hid = a.take<T>(hid_elems * NPL);
        wk.u_plane = (long long)n0;
        wk.u = a.take<T>(n0 * NPL);
        wk.stats = a.take<float2>((size_t)g.ntok[1]);
        ws_bytes = (a.off + 255) / 256 * 256;
    }

    explicit Engine(const Geom& geom, int qkv_a1 = 0, int mlp_mode = 0, int term_plan = 0) : g(geom) {
        attn2 = getenv("SKP_ATTN_V1") == nullptr;
        fused_qa = getenv("SKP_SPLIT_ATTN") == nullptr;
        fused_mlp = (P::NA == 2 && P::NW == 2 && mlp_mode == 0);
        // proj in row-tile form measures the same as the tiled GEMM (0.199 vs 0.197 ms at C = 384, 0.264 vs 0.264 at C = 192: with 16 rows
        // per wave its LDS reads run at 2/3 of the LDS rate): kept behind SKP_RT_PROJ=1, the tiled LayerNorm GEMM stays the default
        rt_proj = (P::NA == 2 && P::NW == 2 && mlp_mode == 0 && getenv("SKP_RT_PROJ") != nullptr);
        fused_block = fused_mlp && getenv("SKP_SPLIT_BLOCK") == nullptr;
        rt_qkv = (std::is_same<P, PrecF16x3>::value && qkv_a1 && mlp_mode == 0);
        plan2 = (std::is_same<P, PrecF16x3>::value && fused_block) ? term_plan : 0;     // the two-term kernel exists for fp16 planes, fused form
        wk.qkv_a1 = qkv_a1;
        params = build_params(g, nullptr);
        plan_prepared(nullptr);
        plan_workspace(nullptr);
    }
    size_t prepared_bytes() const override { return prep_bytes; }
    size_t workspace_bytes() const override { return ws_bytes; }
    void bind(char* prepared, char* workspace) override { plan_prepared(prepared); plan_workspace(workspace); }

    const float* P_(const float* master, const std::string& name) const {
        for (const Param& p : params)
            if (p.name == name) return master + p.offset;
        return nullptr;
    }
    hipError_t copyf(const float* dst, const float* src, size_t n, hipStream_t s) {
        return hipMemcpyAsync(const_cast<float*>(dst), src, n * sizeof(float), hipMemcpyDeviceToDevice, s);
    }
    // blocked = 1 for weights read by the DMA GEMMs, 0 for the register-staged GEMMs (embed, DownSample)
    // perm = 1: "perm8" row order (common.h) -- every GEMM whose epilogue stores 8 consecutive columns per lane
    hipError_t lin(const LinW<T>& l, const float* src, int N, int K, long long sn, long long sk, hipStream_t s, int blocked = 1, int perm = 1) {
        return prep_weight<T, NW>(src, const_cast<T*>(l.w), l.plane, N, K, l.ldw, sn, sk, blocked, perm, s);
    }

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return e_; } while (0)

    hipError_t prepare(const float* m, hipStream_t s) override {
        wk.zrow = zrow;
        CK(hipMemsetAsync(zrow, 0, 4096 * sizeof(T), s));
        CK(copyf(w.mean, P_(m, "norm.mean"), 69, s));
        CK(copyf(w.std, P_(m, "norm.std"), 69, s));
        CK(prep_reciprocal(P_(m, "norm.std"), const_cast<float*>(w.istd), 69, s));
        CK(copyf(w.masks, P_(m, "const_masks"), (size_t)3 * g.n_lat * g.n_lon, s));
        CK(lin(w.embed_u, P_(m, "embed.conv.weight"), 192, 160, 160, 1, s, 0));
        CK(lin(w.embed_s, P_(m, "embed.conv_surface.weight"), 192, 112, 112, 1, s, 0));
        CK(copyf(w.embed_u_b, P_(m, "embed.conv.bias"), 192, s));
        CK(copyf(w.embed_s_b, P_(m, "embed.conv_surface.bias"), 192, s));
        int b = 0;
        for (int layer = 0; layer < 4; ++layer) {
            const int c = layer_dim(layer), heads = layer_heads(layer), res = layer_res(layer);
            for (int i = 0; i < kDepths[layer]; ++i, ++b) {
                const std::string p = "layer" + std::to_string(layer + 1) + ".block" + std::to_string(i) + ".";
                const BlockW<T>& bw = w.blk[b];
                const float *qkv_w = P_(m, p + "attn.qkv.weight"), *qkv_bias = P_(m, p + "attn.qkv.bias");
                if (g.qkv_order) {              // master rows packed (heads, 3, head_dim): one canonical copy, consumed by the preps below (stream order)
                    CK(prep_qkv_rows(qkv_w, qkv_bias, qkv_w_tmp, qkv_b_tmp, c, heads, s));
                    qkv_w = qkv_w_tmp; qkv_bias = qkv_b_tmp;
                }
                CK(lin(bw.qkv, qkv_w, 3 * c, c, c, 1, s));
                CK(lin(bw.proj, P_(m, p + "attn.proj.weight"), c, c, c, 1, s));
                CK(lin(bw.fc1, P_(m, p + "mlp.fc1.weight"), 4 * c, c, c, 1, s));
                CK(lin(bw.fc2, P_(m, p + "mlp.fc2.weight"), c, 4 * c, 4 * c, 1, s));
                if constexpr (P::NA == 2 && P::NW == 2) {
                    if (bw.projf) CK(prep_rowtile_weights<T>(P_(m, p + "attn.proj.weight"), const_cast<T*>(bw.projf), c, c, s));
                    if (bw.qkvf) CK(prep_rowtile_weights<T>(qkv_w, const_cast<T*>(bw.qkvf), 3 * c, c, s));
                    if (bw.qkvh) CK(prep_rowtile_weights<T>(qkv_w, const_cast<T*>(bw.qkvh), 3 * c, c, s, 1));
                    if (bw.w1f) CK(prep_mlp_weights<T>(P_(m, p + "mlp.fc1.weight"), P_(m, p + "mlp.fc2.weight"), const_cast<T*>(bw.w1f), const_cast<T*>(bw.w2f), c, s));
                    if (bw.projh) {
                        CK(prep_rowtile_weights<T>(P_(m, p + "attn.proj.weight"), const_cast<T*>(bw.projh), c, c, s, 1));
                        CK(prep_mlp_weights<T>(P_(m, p + "mlp.fc1.weight"), P_(m, p + "mlp.fc2.weight"), const_cast<T*>(bw.w1h), const_cast<T*>(bw.w2h), c, s, 1));
                    }
                }
                CK(copyf(bw.qkv_b, qkv_bias, 3 * c, s));
                CK(copyf(bw.proj_b, P_(m, p + "attn.proj.bias"), c, s));
                CK(copyf(bw.fc1_b, P_(m, p + "mlp.fc1.bias"), 4 * c, s));
                CK(copyf(bw.fc2_b, P_(m, p + "mlp.fc2.bias"), c, s));
                CK(copyf(bw.n1_g, P_(m, p + "norm1.weight"), c, s));
                CK(copyf(bw.n1_b, P_(m, p + "norm1.bias"), c, s));
                CK(copyf(bw.n2_g, P_(m, p + "norm2.weight"), c, s));
                CK(copyf(bw.n2_b, P_(m, p + "norm2.bias"), c, s));
                if (attn2) CK(prep_bias_compact(P_(m, p + "attn.bias_table"), const_cast<f16*>(bw.bias_cmp), g.types[res], heads, g.nH[res], (i & 1) ? g.roll_sign : 0, g.mask_value, s, g.bias_transposed));
                else CK(prep_bias_expand(P_(m, p + "attn.bias_table"), const_cast<f16*>(bw.bias_exp), g.types[res], heads, g.nH[res], (i & 1) ? g.roll_sign : 0, g.mask_value, s, g.bias_transposed));
            }
        }
        CK(copyf(w.down_g, P_(m, "down.norm.weight"), 768, s));
        CK(copyf(w.down_b, P_(m, "down.norm.bias"), 768, s));
        CK(lin(w.down, P_(m, "down.linear.weight"), 384, 768, 768, 1, s, 0));
        CK(lin(w.up1, P_(m, "up.linear1.weight"), 768, 384, 384, 1, s));
        CK(lin(w.up2, P_(m, "up.linear2.weight"), 192, 192, 192, 1, s));
        CK(copyf(w.up_g, P_(m, "up.norm.weight"), 192, s));
        CK(copyf(w.up_b, P_(m, "up.norm.bias"), 192, s));
        // ConvTranspose weights are [K = 384][N]; the GEMM wants [N][K]
        CK(lin(w.rec_u, P_(m, "recover.conv.weight"), 160, 384, 1, 160, s, 1, 0));     // EpRecover: 4 = one float4 of longitudes
        CK(lin(w.rec_s, P_(m, "recover.conv_surface.weight"), 64, 384, 1, 64, s, 1, 0));
        CK(copyf(w.rec_u_b, P_(m, "recover.conv.bias"), 5, s));
        CK(copyf(w.rec_s_b, P_(m, "recover.conv_surface.bias"), 4, s));
        const int H[2] = {g.H1, g.H2}, W[2] = {g.W1, g.W2};
        for (int r = 0; r < 2; ++r)
            for (int roll = 0; roll < 2; ++roll)
            {
                CK(prep_window_index(const_cast<int*>(w.widx[r][roll]), g.Z, H[r], W[r], g.Hp[r], g.top[r], roll ? g.roll_sign : 0, s, g.surface_last));
                CK(prep_window_inverse(w.widx[r][roll], g.mwin[r], const_cast<int*>(w.winv[r][roll]), s));
            }
        return hipSuccess;
    }

    static int block_index(int layer0, int i) {
        int b = 0;
        for (int l = 0; l < layer0; ++l) b += kDepths[l];
        return b + i;
    }

    // one EarthSpecificBlock on a residual stream (hi/lo planes)
    hipError_t block_planes(int layer0, int i, T* xs, hipStream_t s) {
        const int res = layer_res(layer0), C = layer_dim(layer0), heads = layer_heads(layer0);
        const BlockW<T>& bw = w.blk[block_index(layer0, i)];
        const int* widx = w.widx[res][i & 1];
        const int o = res == 0 ? 0 : 5;
        bool fused = false;
        if constexpr (std::is_same<P, PrecF16x3>::value) {
            // QKV + attention in one launch: the row-tile QKV forms (stream hi plane; one weight plane, or hi / lo at C = 192) with the compact bias
            if (qa_fused(bw, res)) {
                mark(C_ATTN0 + o, s);
                CK(op_qkv_attention(g, bw, widx, res, xs, wk, s, block_one(layer0) ? 1 : 2));
                fused = true;
            }
        }
        if (!fused) {
            mark(C_QKV0 + o, s);
            if constexpr (std::is_same<P, PrecF16x3>::value) {
                if (rt_qkv) CK(op_qkv_rowtile(g, bw, widx, res, xs, wk, s));
                else CK((op_qkv<P>(g, bw, widx, res, xs, wk, s)));
            } else {
                CK((op_qkv<P>(g, bw, widx, res, xs, wk, s)));
            }
            mark(C_ATTN0 + o, s);
            AttnArgs<P> a{wk.q, wk.k, wk.vt, wk.qkv_plane, bw.bias_exp, bw.bias_cmp, wk.ao, wk.ao_plane, C, g.nwin[res], g.nW[res], heads};
            a.out_planes = block_one(layer0) ? 1 : 0;           // a one-term block kernel reads the hi plane only
            CK(launch_attention<P>(a, s));
        }
        if constexpr (std::is_same<P, PrecF16x3>::value) {
            if (two_term(layer0)) {               // ... with the weights as one fp16 plane: two MFMA terms, or one
                mark(C_FC1_0 + o, s);
                CK(op_proj_mlp2(g, bw, w.winv[res][i & 1], res, xs, wk, s, block_one(layer0)));
                mark(-1, s);
                return hipSuccess;
            }
        }
        if constexpr (P::NA == 2 && P::NW == 2) {
            if (fused_block) {                    // everything after the attention in one kernel; timed under "mlp"
                mark(C_FC1_0 + o, s);
                CK((op_proj_mlp_fused<P>(g, bw, w.winv[res][i & 1], res, xs, wk, s)));
                mark(-1, s);
                return hipSuccess;
            }
        }
        mark(C_PROJ0 + o, s);
        if constexpr (P::NA == 2 && P::NW == 2) {
            if (rt_proj) CK((op_proj_rowtile<P>(g, bw, widx, res, xs, wk, s)));
            else CK((op_proj<P>(g, bw, widx, res, xs, wk, s)));
        } else {
            CK((op_proj<P>(g, bw, widx, res, xs, wk, s)));
        }
        if constexpr (P::NA == 2 && P::NW == 2) {
            if (fused_mlp) {                      // timed under the fc1 category ("mlp" when fused); fc2 has no launch of its own
                mark(C_FC1_0 + o, s);
                CK((op_mlp_fused<P>(g, bw, res, xs, wk, s)));
                mark(-1, s);
                return hipSuccess;
            }
        }
        mark(C_FC1_0 + o, s);
        CK((op_fc1<P>(g, bw, res, xs, wk, s)));
        mark(C_FC2_0 + o, s);
        CK((op_fc2<P>(g, bw, res, xs, wk, s)));
        mark(-1, s);
        return hipSuccess;
    }
    // stage-level API: the caller hands / receives fp32 row-major tensors; they are converted to / from planes here
    hipError_t to_planes(const float* x, T* xs, int res, hipStream_t s) {
        return split_planes<T, 2>(x, xs, wk.xs_plane[res], wk.xs_plane[res], res == 0 ? 192 : 384, s);
    }
    hipError_t from_planes(const T* xs, float* x, int res, hipStream_t s) {
        return merge_planes<T>(xs, wk.xs_plane[res], x, wk.xs_plane[res], res == 0 ? 192 : 384, s);
    }
    hipError_t block(int layer0, int i, float* x, hipStream_t s) override {
        const int res = layer_res(layer0);
        T* xs = res == 0 ? wk.X1s : wk.X2s;
        CK(to_planes(x, xs, res, s));
        CK(block_planes(layer0, i, xs, s));
        return from_planes(xs, x, res, s);
    }
    hipError_t embed(const float* in, float* x1, hipStream_t s) override {
        CK((op_embed<P>(g, w, in, wk.X1s, wk, s)));
        return from_planes(wk.X1s, x1, 0, s);
    }
    hipError_t down(const float* x1, float* x2, hipStream_t s) override {
        CK(to_planes(x1, wk.X1s, 0, s));
        CK((op_down<P>(g, w, wk.X1s, wk.X2s, wk, s)));
        return from_planes(wk.X2s, x2, 1, s);
    }
    hipError_t up(const float* x2, float* x4, hipStream_t s) override {
        CK(to_planes(x2, wk.X2s, 1, s));
        CK((op_up<P>(g, w, wk.X2s, wk.X4s, wk, s)));
        return from_planes(wk.X4s, x4, 0, s);
    }
    hipError_t recover(const float* skip, const float* x4, float* out, hipStream_t s) override {
        CK(to_planes(skip, wk.X1s, 0, s));
        CK(to_planes(x4, wk.X4s, 0, s));
        return op_recover<P>(g, w, wk.X1s, wk.X4s, out, wk, s);
    }

    // ---- calibration of the term plan ---- //
    // A GEMM that runs with its weights as ONE fp16 plane drops A x (W - fp16(W)).  Over the tokens of a state that term has a mean and a
    // spread; the mean is a constant row vector and belongs in the bias.  calibrate() runs one step on `in` through the three-term kernels
    // (the tiled path: every operand reaches HBM there), takes the column means of the operand of each GEMM that the plan runs short, and
    // adds (W - fp16(W)) x mean to that Linear's bias.  Idempotent: the biases are re-read from the master first.
    hipError_t calib_block(const float* m, int layer0, int i, T* xs, hipStream_t s) {
        const int res = layer_res(layer0), C = layer_dim(layer0), heads = layer_heads(layer0);
        const BlockW<T>& bw = w.blk[block_index(layer0, i)];
        const int* widx = w.widx[res][i & 1];
        const int ntok = g.ntok[res];
        const float inv = 1.0f / (float)ntok;
        // per-workgroup partial sums: Q / K / V are dead at every point a column sum is taken (before the QKV GEMM, after the attention)
        float* scratch = reinterpret_cast<float*>(wk.q);
        if (colsum_scratch_floats(ntok, 4 * C) * sizeof(float) > 3 * q_elems * sizeof(f16)) return hipErrorInvalidValue;
        const std::string p = "layer" + std::to_string(layer0 + 1) + ".block" + std::to_string(i) + ".";
        const bool t2 = two_term(layer0), q1 = qkv_one(layer0);
        const int npl = block_one(layer0) ? 1 : 2;                    // planes of the activation operands the layer's block GEMMs read
        const float *qkv_w = P_(m, p + "attn.qkv.weight"), *qkv_bias = P_(m, p + "attn.qkv.bias");
        if (g.qkv_order) { CK(prep_qkv_rows(qkv_w, qkv_bias, qkv_w_tmp, qkv_b_tmp, C, heads, s)); qkv_w = qkv_w_tmp; qkv_bias = qkv_b_tmp; }
        CK(copyf(bw.qkv_b, qkv_bias, 3 * C, s));
        CK(copyf(bw.proj_b, P_(m, p + "attn.proj.bias"), C, s));
        CK(copyf(bw.fc1_b, P_(m, p + "mlp.fc1.bias"), 4 * C, s));
        CK(copyf(bw.fc2_b, P_(m, p + "mlp.fc2.bias"), C, s));
        if (q1) CK(colsum_planes<T>(xs, 0, 1, nullptr, ntok, C, scratch, cal_sum, s));                       // one-term QKV reads the stream's hi plane
        CK((op_qkv<P>(g, bw, widx, res, xs, wk, s)));
        if (q1) CK(bias_fold(qkv_w, cal_sum, inv, const_cast<float*>(bw.qkv_b), 3 * C, C, s));
        AttnArgs<P> a{wk.q, wk.k, wk.vt, wk.qkv_plane, bw.bias_exp, bw.bias_cmp, wk.ao, wk.ao_plane, C, g.nwin[res], g.nW[res], heads};
        CK(launch_attention<P>(a, s));
        if (t2) CK(colsum_planes<T>(wk.ao, wk.ao_plane, npl, w.winv[res][i & 1], ntok, C, scratch, cal_sum, s));   // window rows of the real tokens
        CK((op_proj<P>(g, bw, widx, res, xs, wk, s)));
        if (t2) {
            CK(bias_fold(P_(m, p + "attn.proj.weight"), cal_sum, inv, const_cast<float*>(bw.proj_b), C, C, s));
            CK(colsum_planes<T>(xs, wk.xs_plane[res], npl, nullptr, ntok, C, scratch, cal_sum, s));             // the mid-block stream
        }
        CK((op_fc1<P>(g, bw, res, xs, wk, s)));
        if (t2) {
            CK(bias_fold(P_(m, p + "mlp.fc1.weight"), cal_sum, inv, const_cast<float*>(bw.fc1_b), 4 * C, C, s));
            CK(colsum_planes<T>(wk.hid, wk.hid_plane, npl, nullptr, ntok, 4 * C, scratch, cal_sum, s));
        }
        CK((op_fc2<P>(g, bw, res, xs, wk, s)));
        if (t2) CK(bias_fold(P_(m, p + "mlp.fc2.weight"), cal_sum, inv, const_cast<float*>(bw.fc2_b), C, 4 * C, s));
        return hipSuccess;
    }
    hipError_t master_biases(const float* m, hipStream_t s) {
        int b = 0;
        for (int layer = 0; layer < 4; ++layer)
            for (int i = 0; i < kDepths[layer]; ++i, ++b) {
                const int C = layer_dim(layer), heads = layer_heads(layer);
                const std::string p = "layer" + std::to_string(layer + 1) + ".block" + std::to_string(i) + ".";
                const BlockW<T>& bw = w.blk[b];
                const float *qkv_w = P_(m, p + "attn.qkv.weight"), *qkv_bias = P_(m, p + "attn.qkv.bias");
                if (g.qkv_order) { CK(prep_qkv_rows(qkv_w, qkv_bias, qkv_w_tmp, qkv_b_tmp, C, heads, s)); qkv_bias = qkv_b_tmp; }
                CK(copyf(bw.qkv_b, qkv_bias, 3 * C, s));
                CK(copyf(bw.proj_b, P_(m, p + "attn.proj.bias"), C, s));
                CK(copyf(bw.fc1_b, P_(m, p + "mlp.fc1.bias"), 4 * C, s));
                CK(copyf(bw.fc2_b, P_(m, p + "mlp.fc2.bias"), C, s));
            }
        return hipSuccess;
    }
    hipError_t calibrate(const float* m, const float* in, hipStream_t s) override {
        if (plan2 == 0) return hipSuccess;               // nothing runs short
        if (!in) return master_biases(m, s);             // no state: back to the uncalibrated plan
        if constexpr (std::is_same<P, PrecF16x3>::value) {
            CK((op_embed<P>(g, w, in, wk.X1s, wk, s)));
            for (int i = 0; i < kDepths[0]; ++i) CK(calib_block(m, 0, i, wk.X1s, s));
            CK((op_down<P>(g, w, wk.X1s, wk.X2s, wk, s)));
            for (int i = 0; i < kDepths[1]; ++i) CK(calib_block(m, 1, i, wk.X2s, s));
            for (int i = 0; i < kDepths[2]; ++i) CK(calib_block(m, 2, i, wk.X2s, s));
            CK((op_up<P>(g, w, wk.X2s, wk.X4s, wk, s)));
            for (int i = 0; i < kDepths[3]; ++i) CK(calib_block(m, 3, i, wk.X4s, s));
        }
        return hipSuccess;
    }

    hipError_t step(const float* in, float* out, hipStream_t s) override {
        mark(C_EMBED, s);
        CK((op_embed<P>(g, w, in, wk.X1s, wk, s)));
        for (int i = 0; i < kDepths[0]; ++i) CK(block_planes(0, i, wk.X1s, s));
        mark(C_DOWN, s);
        CK((op_down<P>(g, w, wk.X1s, wk.X2s, wk, s)));
        for (int i = 0; i < kDepths[1]; ++i) CK(block_planes(1, i, wk.X2s, s));
        for (int i = 0; i < kDepths[2]; ++i) CK(block_planes(2, i, wk.X2s, s));
        mark(C_UP, s);
        CK((op_up<P>(g, w, wk.X2s, wk.X4s, wk, s)));
        for (int i = 0; i < kDepths[3]; ++i) CK(block_planes(3, i, wk.X4s, s));
        mark(C_RECOVER, s);
        CK((op_recover<P>(g, w, wk.X1s, wk.X4s, out, wk, s)));
        mark(-1, s);
        return hipSuccess;
    }
#undef CK

    bool debug(const std::string& n, void** p, size_t* bytes) override {
        auto set = [&](const void* ptr, size_t b) { *p = const_cast<void*>(ptr); *bytes = b; return true; };
        if (n == "q") return set(wk.q, q_elems * sizeof(f16));
        if (n == "k") return set(wk.k, q_elems * sizeof(f16));
        if (n == "vt") return set(wk.vt, q_elems * sizeof(f16));
        if (n == "ao") return set(wk.ao, ao_elems * NPL * sizeof(T));
        if (n == "hid") return set(wk.hid, hid_elems * NPL * sizeof(T));
        if (n == "u") return set(wk.u, (size_t)g.ntok[0] * 192 * NPL * sizeof(T));
        if (n == "x1") return set(wk.X1s, (size_t)g.ntok[0] * 192 * 2 * sizeof(T));
        if (n == "x2") return set(wk.X2s, (size_t)g.ntok[1] * 384 * 2 * sizeof(T));
        if (n == "x4") return set(wk.X4s, (size_t)g.ntok[0] * 192 * 2 * sizeof(T));
        if (n.rfind("widx", 0) == 0 && n.size() == 6) {
            const int r = n[4] - '0', roll = n[5] - '0';
            if (r < 0 || r > 1 || roll < 0 || roll > 1) return false;
            return set(w.widx[r][roll], (size_t)g.mwin[r] * 4);
        }
        if (n.rfind("bias_exp", 0) == 0) {
            const int b = atoi(n.c_str() + 8);
            if (b < 0 || b > 15) return false;
            return set(attn2 ? w.blk[b].bias_cmp : w.blk[b].bias_exp, bias_exp_elems[b] * sizeof(f16));
        }
        return false;
    }
};

IEngine* make_engine(const skpangu_config& cfg, const Geom& g) {
    switch (cfg.precision) {
        case SKPANGU_PREC_BF16X3: return new Engine<PrecBF16x3>(g, 0, cfg.mlp_mode);
        case SKPANGU_PREC_F16X3: return new Engine<PrecF16x3>(g, 0, cfg.mlp_mode, cfg.term_plan);
        case SKPANGU_PREC_F16X3_Q: return new Engine<PrecF16x3>(g, 1, cfg.mlp_mode, cfg.term_plan);
        default: return nullptr;
    }
}

}  // namespace

struct skpangu_ctx {
    skpangu_config cfg;
    Geom g;
    IEngine* eng;
    bool prepared;
};

extern "C" {

int skpangu_abi_version(void) { return SKPANGU_ABI_VERSION; }

const char* skpangu_error_string(int code) {
    switch (code) {
        case 0: return "success";
        case SKPANGU_E_ARG: return "bad argument or unsupported geometry";
        case SKPANGU_E_SIZE: return "caller buffer too small";
        case SKPANGU_E_STATE: return "context not prepared";
        case SKPANGU_E_NOTFOUND: return "not found";
        default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
    }
}

int skpangu_query_sizes(const skpangu_config* cfg, skpangu_sizes* out) {
    if (!cfg || !out) return SKPANGU_E_ARG;
    Geom g;
    if (!make_geom(*cfg, g)) return SKPANGU_E_ARG;
    IEngine* e = make_engine(*cfg, g);
    if (!e) return SKPANGU_E_ARG;
    long long total = 0;
    std::vector<Param> p = build_params(g, &total);
    out->master_floats = total;
    out->prepared_bytes = e->prepared_bytes();
    out->workspace_bytes = e->workspace_bytes();
    out->state_floats = 69LL * g.n_lat * g.n_lon;
    out->n_params = (int)p.size();
    delete e;
    return 0;
}

int skpangu_param_info(const skpangu_config* cfg, int index, char* name, size_t name_cap, long long* offset, int* ndim, long long shape[6]) {
    if (!cfg) return SKPANGU_E_ARG;
    Geom g;
    if (!make_geom(*cfg, g)) return SKPANGU_E_ARG;
    std::vector<Param> p = build_params(g, nullptr);
    if (index < 0 || index >= (int)p.size()) return SKPANGU_E_NOTFOUND;
    const Param& q = p[index];
    if (name && name_cap) { std::strncpy(name, q.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
    if (offset) *offset = q.offset;
    if (ndim) *ndim = (int)q.shape.size();
    if (shape) for (size_t i = 0; i < q.shape.size() && i < 6; ++i) shape[i] = q.shape[i];
    return 0;
}

int skpangu_create(const skpangu_config* cfg, void* prepared_dev, size_t prepared_bytes, void* workspace_dev, size_t workspace_bytes, skpangu_ctx** out) {
    if (!cfg || !out || !prepared_dev || !workspace_dev) return SKPANGU_E_ARG;
    Geom g;
    if (!make_geom(*cfg, g)) return SKPANGU_E_ARG;
    IEngine* e = make_engine(*cfg, g);
    if (!e) return SKPANGU_E_ARG;
    if (prepared_bytes < e->prepared_bytes() || workspace_bytes < e->workspace_bytes()) { delete e; return SKPANGU_E_SIZE; }
    if (((uintptr_t)prepared_dev & 255) || ((uintptr_t)workspace_dev & 255)) { delete e; return SKPANGU_E_ARG; }
    e->bind((char*)prepared_dev, (char*)workspace_dev);
    skpangu_ctx* c = new skpangu_ctx{*cfg, g, e, false};
    *out = c;
    return 0;
}

void skpangu_destroy(skpangu_ctx* ctx) {
    if (!ctx) return;
    delete ctx->eng;
    delete ctx;
}

int skpangu_prepare(skpangu_ctx* ctx, const float* master_dev, void* stream) {
    if (!ctx || !master_dev) return SKPANGU_E_ARG;
    const int e = (int)ctx->eng->prepare(master_dev, (hipStream_t)stream);
    if (e == 0) ctx->prepared = true;
    return e;
}

#define NEED_PREPARED() do { if (!ctx) return SKPANGU_E_ARG; if (!ctx->prepared) return SKPANGU_E_STATE; } while (0)

int skpangu_calibrate(skpangu_ctx* ctx, const float* master_dev, const float* state_in, void* stream) {
    NEED_PREPARED();
    if (!master_dev) return SKPANGU_E_ARG;
    return (int)ctx->eng->calibrate(master_dev, state_in, (hipStream_t)stream);
}

int skpangu_step(skpangu_ctx* ctx, const float* in, float* out, void* stream) {
    NEED_PREPARED();
    if (!in || !out) return SKPANGU_E_ARG;
    return (int)ctx->eng->step(in, out, (hipStream_t)stream);
}
int skpangu_patch_embed(skpangu_ctx* ctx, const float* in, float* x1, void* stream) {
    NEED_PREPARED();
    if (!in || !x1) return SKPANGU_E_ARG;
    return (int)ctx->eng->embed(in, x1, (hipStream_t)stream);
}
int skpangu_block(skpangu_ctx* ctx, int layer, int block, float* x, void* stream) {
    NEED_PREPARED();
    if (!x || layer < 1 || layer > 4 || block < 0 || block >= kDepths[layer - 1]) return SKPANGU_E_ARG;
    return (int)ctx->eng->block(layer - 1, block, x, (hipStream_t)stream);
}
int skpangu_downsample(skpangu_ctx* ctx, const float* x1, float* x2, void* stream) {
    NEED_PREPARED();
    if (!x1 || !x2) return SKPANGU_E_ARG;
    return (int)ctx->eng->down(x1, x2, (hipStream_t)stream);
}
int skpangu_upsample(skpangu_ctx* ctx, const float* x2, float* x4, void* stream) {
    NEED_PREPARED();
    if (!x2 || !x4) return SKPANGU_E_ARG;
    return (int)ctx->eng->up(x2, x4, (hipStream_t)stream);
}
int skpangu_patch_recover(skpangu_ctx* ctx, const float* skip, const float* x4, float* out, void* stream) {
    NEED_PREPARED();
    if (!skip || !x4 || !out) return SKPANGU_E_ARG;
    return (int)ctx->eng->recover(skip, x4, out, (hipStream_t)stream);
}
int skpangu_profile(skpangu_ctx* ctx, int enable) {
    if (!ctx) return SKPANGU_E_ARG;
    ctx->eng->profile(enable != 0);
    return 0;
}
int skpangu_profile_read(skpangu_ctx* ctx, skpangu_stage_stat* out, int cap, int* n) {
    if (!ctx || !out || !n || cap <= 0) return SKPANGU_E_ARG;
    return (int)ctx->eng->profile_read(out, cap, n);
}
int skpangu_debug_buffer(skpangu_ctx* ctx, const char* name, void** ptr, size_t* bytes) {
    if (!ctx || !name || !ptr || !bytes) return SKPANGU_E_ARG;
    return ctx->eng->debug(name, ptr, bytes) ? 0 : SKPANGU_E_NOTFOUND;
}

}  // extern "C"
