// Everything of an EarthSpecificBlock that follows the attention as ONE kernel, WIDE row tiles: one wave per SIMD owns 32 (C = 384) or 64
// (C = 192) stream tokens and the whole 512-register file of its SIMD.
//
// Why.  In the row-tile kernels only WEIGHTS cross LDS, and every wave reads every weight fragment.  With 16 rows per wave (fused_block2.hip:
// two waves per SIMD, 256 registers each) a ds_read_b128 feeds one MFMA per term, so at one term (A_hi W_hi) the LDS pipe is as busy as the
// matrix pipe (4 clocks per CU for the read, 16 clocks per SIMD for the MFMA) and at two terms half as busy, on top of twice the LDS-DMA
// bytes -- the kernel measured 0.70 ms per launch at C = 384 with its matrix pipes 48 % busy and barely gained from dropping a term.  Here a
// fragment read feeds FM = 2 (C = 384) / 4 (C = 192) row groups: half the LDS reads and half the L2 -> LDS weight traffic per token, four
// independent accumulator chains in every loop (a v_mfma that accumulates into the previous one's result issues ~47 clocks after it), and the
// registers of a second wave are spent on rows instead.
//
// Structure.  Persistent workgroups (grid = CUs), 4 waves, tiles of 4 x FM x 16 stream tokens.  The weights of a tile are ONE stream of
// `units` of two LDS slots -- NPB / 2 pairs of projection blocks, the unit {W1(0)}, then {W1(j + 1), W2(j)} for the 4C / 32 hidden chunks --
// through a ring of three units: at the top of unit u every wave waits for its own pieces of unit u (requested two units earlier:
// s_waitcnt vmcnt(PIECES), unit u + 1 may still be in flight), ONE barrier, and unit u + 2 is requested into the ring position unit u - 1 has
// just left.  The stream wraps around into the next tile, so the weight DMA never drains between tiles.  A unit pairs fc2's chunk j with
// fc1's chunk j + 1 because that is what one interval of the loop touches: PIPE runs  | fc1(j + 1) with GELU(j) spliced between its MFMAs |
// fc2(j) |  (one wave per SIMD: nothing else would run under the VALU phase), the serial order is  | GELU(j) | fc2(j) | fc1(j + 1) |.
// Row-tile algebra, fragment orders and epilogues are fused_block2.hip's: the attention rows are gathered through the inverse window table as
// B-operand fragments, x_mid = x + LayerNorm(proj) becomes the MLP's operand in registers (perm8), fc1 -> GELU -> fc2 without leaving the
// registers, the stream is read once and written once.  gfx950 only.
#include <cstdlib>
#include "blockrow.h"
#include "launchers.h"

namespace skp {

template <int C_, int FM_, bool ONE_, bool PIPE_, int RD_ = 2, int PROBE_ = 0>
struct WideShape {
    // timing probes (tools/micro/wide_probe.hip only; results are wrong): 1 no GELU polynomial, 2 one fragment pair read per phase, 4 no weight
    // DMA in the MLP loop, 8 no barrier / DMA wait in the MLP loop, 16 no row gathers / stores, 32 no MFMAs in the MLP loop
    static constexpr int PROBE = PROBE_;
    // ONE: the activation operands as one fp16 plane (A_hi W); otherwise two terms (A_hi W + A_lo W).  The weights are one plane either way.
    static constexpr bool ONE = ONE_, PIPE = PIPE_;
    static constexpr int C = C_, FM = FM_, NWAVES = 4, THREADS = 64 * NWAVES, RD = RD_;
    static constexpr int KS = C / 32, CF = C / 16, HID = 4 * C, NCH = HID / 32, NPB = C / 32, BM = NWAVES * FM * 16;
    static constexpr int SLOT_KIB = KS * 2, SLOT = SLOT_KIB * 1024, UNIT = 2 * SLOT, NRING = 3;
    static constexpr int NPU = NPB / 2, NU = NPU + 1 + NCH;       // units of a tile: projection block pairs, {W1(0)}, {W1(j + 1), W2(j)}
    static constexpr int HALF = SLOT_KIB / NWAVES, PIECES = 2 * HALF;   // LDS-DMA instructions per wave: per slot, per unit
    static constexpr int T_PB = 0, T_G1 = C, T_E1 = 2 * C, T_B1 = 3 * C, T_B2 = 3 * C + HID, T_G2 = T_B2 + C, T_E2 = T_G2 + C;
    static constexpr int TAB = T_E2 + C;
    static constexpr int SMEM = NRING * UNIT + TAB * 4;
    // PIPE: the 4 FM packed GELU evaluations of a chunk go EPK per k-step into fc1's first NEK k-steps, the fp16 conversions into the rest
    static constexpr int NE = 4 * FM, EPK = (NE + KS - 3) / (KS - 2), NEK = (NE + EPK - 1) / EPK, TPK = (FM + KS - NEK - 1) / (KS - NEK);
    static_assert(SLOT_KIB % NWAVES == 0 && NPB % 2 == 0 && NCH >= 3, "pieces per wave and slot; whole projection units");
    static_assert(NEK < KS && TPK * (KS - NEK) >= FM, "PIPE: the GELU pieces fit fc1's k-steps");
    static_assert(SMEM <= 160 * 1024, "LDS");
};

template <class T, class S>
__global__ void __launch_bounds__(S::THREADS) __attribute__((amdgpu_waves_per_eu(1, 1)))
proj_mlp_wide_kernel(const Block2Args<T> a, const int ntiles) {
    constexpr int C = S::C, FM = S::FM, KS = S::KS, CF = S::CF, NCH = S::NCH, NPU = S::NPU, NU = S::NU, NWAVES = S::NWAVES, RD = S::RD;
    constexpr bool P_GELU = S::PROBE & 1, P_LDS = S::PROBE & 2, P_DMA = S::PROBE & 4, P_BAR = S::PROBE & 8, P_IO = S::PROBE & 16, P_MFMA = S::PROBE & 32;
    typedef typename OpT<T>::v8 v8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* tab = reinterpret_cast<float*>(smem + S::NRING * S::UNIT);
    const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, g = lane >> 4;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds_base = (unsigned)(size_t)smem;
    const char* lrd = smem + lane * 16;                          // a lane's 16 bytes of every fragment

    // unit u of a tile -> ring position rp: two slots of SLOT_KIB pieces, piece q of a slot by wave q % NWAVES.  The half-empty units at the
    // ends of the MLP ({W1(0)}, {W2(NCH - 1)}) fetch a neighbouring chunk into the unused slot: every unit is PIECES loads per wave.
    auto dma_unit = [&](int u, int rp) {
        const T *s0, *s1;
        if (u < NPU) { s0 = a.projh + ((long long)(2 * u) * S::SLOT_KIB << 9); s1 = s0 + (S::SLOT_KIB << 9); }
        else {
            const int j = u - NPU - 1;                           // {W1(j + 1), W2(j)}
            s0 = a.w1h + ((long long)(j + 1 < NCH ? j + 1 : j) * S::SLOT_KIB << 9);
            s1 = a.w2h + ((long long)(j < 0 ? 0 : j) * S::SLOT_KIB << 9);
        }
        s0 += lane * 8 + (wave << 9); s1 += lane * 8 + (wave << 9);
        const unsigned dst = lds_base + (unsigned)(rp * S::UNIT) + (unsigned)(wave << 10);
#pragma unroll
        for (int i = 0; i < S::HALF; ++i) glds16(s0 + (i * NWAVES << 9), dst + (unsigned)(i * NWAVES << 10));
#pragma unroll
        for (int i = 0; i < S::HALF; ++i) glds16(s1 + (i * NWAVES << 9), dst + (unsigned)(S::SLOT + (i * NWAVES << 10)));
    };

    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    dma_unit(0, 0);
    dma_unit(1, 1);
    for (int i = tid; i < C; i += S::THREADS) {
        tab[S::T_PB + i] = a.proj_b[i]; tab[S::T_G1 + i] = a.g1[i]; tab[S::T_E1 + i] = a.e1[i];
        tab[S::T_B2 + i] = a.b2[i]; tab[S::T_G2 + i] = a.g2[i]; tab[S::T_E2 + i] = a.e2[i];
    }
    for (int i = tid; i < S::HID; i += S::THREADS) tab[S::T_B1 + i] = a.b1[i];

    // Top of unit u (ring position rp): its weights landed, every wave is done with unit u - 1; unit u + 2 (of this tile or the next) goes into
    // the position unit u - 1 has left.  `pending`: the previous top requested a unit and nothing else has touched vector memory since -- only
    // then may PIECES loads stay in flight; returns whether this top requested one.
    bool has_next = false;
    int rp = 0;                                                   // ring position of the current unit, running across tiles
    auto top = [&](int u, bool pending) {
        const bool mlp = u > NPU;
        if (!(P_BAR && mlp)) {
            if (pending) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(S::PIECES) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        const int rp2 = rp == 0 ? 2 : rp - 1;                    // (rp + 2) % 3
        bool req = true;
        if (P_DMA && mlp && u + 2 < NU) return false;
        if (u + 2 < NU) dma_unit(u + 2, rp2);
        else if (has_next) dma_unit(u + 2 - NU, rp2);
        else req = false;
        return req;
    };
    auto advance = [&] { rp = rp == 2 ? 0 : rp + 1; };

    for (; tile < ntiles; tile += gridDim.x) {
        has_next = tile + (int)gridDim.x < ntiles;
        // the attention rows of the wave's tokens as B-operand fragments (gathered: 16 bytes per lane, 64 bytes per row and k-step)
        const long long rb0 = (long long)tile * (S::BM / 16) + wave * FM;
        v8 xh[FM][KS], xl[FM][KS];
        bool live[FM];
#pragma unroll
        for (int t = 0; t < FM; ++t) {
            live[t] = (rb0 + t) * 16 < a.M;
            const int src = live[t] && !P_IO ? a.winv[(rb0 + t) * 16 + l15] : 0;
            const T* p = a.ao + blk_off(src, g * 8, C);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                if constexpr (P_IO) { xh[t][ks] = v8{}; xl[t][ks] = v8{}; xh[t][ks][0] = (T)(float)lane; continue; }
                xh[t][ks] = *reinterpret_cast<const v8*>(p + (ks << 9));
                if constexpr (S::ONE) xl[t][ks] = v8{};              // the lo plane of the attention output is not read at all
                else xl[t][ks] = *reinterpret_cast<const v8*>(p + (ks << 9) + a.ao_plane);
            }
        }
#pragma unroll
        for (int t = 0; t < FM; ++t) {              // consumed once here: hipcc's vmcnt waits for these loads sit BEFORE any later DMA request
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                asm volatile("" : "+v"(xh[t][ks]));
                if constexpr (!S::ONE) asm volatile("" : "+v"(xl[t][ks]));
            }
        }

        f32x4 yacc[FM][CF];
#pragma unroll
        for (int t = 0; t < FM; ++t)
#pragma unroll
            for (int c = 0; c < CF; ++c) yacc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};

        // ---- phase 1: projection, two blocks of 32 output columns per unit ------------------------------------------------------------ //
        bool pending = false;                           // the gathers above (and the previous tile's stores) went through vector memory
#pragma unroll
        for (int u = 0; u < NPU; ++u) {
            pending = top(u, pending);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int j = 2 * u + h;
                sk_stream<KS, RD>(lrd + rp * S::UNIT + h * S::SLOT, [&](int ks, const uint4& w0, const uint4& w1) {
#pragma unroll
                    for (int t = 0; t < FM; ++t) yacc[t][2 * j] = OpT<T>::mfma(as_v8<T>(w0), xh[t][ks], yacc[t][2 * j]);
#pragma unroll
                    for (int t = 0; t < FM; ++t) yacc[t][2 * j + 1] = OpT<T>::mfma(as_v8<T>(w1), xh[t][ks], yacc[t][2 * j + 1]);
                    if constexpr (!S::ONE) {
#pragma unroll
                        for (int t = 0; t < FM; ++t) yacc[t][2 * j] = OpT<T>::mfma(as_v8<T>(w0), xl[t][ks], yacc[t][2 * j]);
#pragma unroll
                        for (int t = 0; t < FM; ++t) yacc[t][2 * j + 1] = OpT<T>::mfma(as_v8<T>(w1), xl[t][ks], yacc[t][2 * j + 1]);
                    }
                });
            }
            advance();
        }

        // ---- between: x_mid = x + LayerNorm(yacc + bias) -> the MLP's input fragments (the attention fragments are dead) -------------- //
#pragma unroll
        for (int t = 0; t < FM; ++t) {
            const T* old = a.xs + ((live[t] ? rb0 + t : 0) * KS << 9) + l15 * 32 + g * 8;
            v8 oh[KS], ol[KS];
#pragma unroll
            for (int bp = 0; bp < KS; ++bp) {
                if constexpr (P_IO) { oh[bp] = xh[t][bp]; ol[bp] = xl[t][bp]; continue; }
                oh[bp] = *reinterpret_cast<const v8*>(old + (bp << 9));
                ol[bp] = *reinterpret_cast<const v8*>(old + (bp << 9) + a.xs_plane);
            }
            float s = 0.f;
#pragma unroll
            for (int bp = 0; bp < KS; ++bp) {
                const int n = 32 * bp + 8 * g;
                const float4 b0 = *reinterpret_cast<const float4*>(tab + S::T_PB + n), b1 = *reinterpret_cast<const float4*>(tab + S::T_PB + n + 4);
                add8(yacc[t][2 * bp], yacc[t][2 * bp + 1], b0, b1);
#pragma unroll
                for (int r = 0; r < 4; ++r) s += yacc[t][2 * bp][r] + yacc[t][2 * bp + 1][r];
            }
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            const float mean = s * (1.0f / C);
            float q = 0.f;
#pragma unroll
            for (int c = 0; c < CF; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float d = yacc[t][c][r] - mean; q += d * d; }
            q += __shfl_xor(q, 16);
            q += __shfl_xor(q, 32);
            const float rstd = rsqrtf(q * (1.0f / C) + a.eps);
#pragma unroll
            for (int bp = 0; bp < KS; ++bp) {
                const int n = 32 * bp + 8 * g;
                const float4 g0 = *reinterpret_cast<const float4*>(tab + S::T_G1 + n), g1 = *reinterpret_cast<const float4*>(tab + S::T_G1 + n + 4);
                const float4 e0 = *reinterpret_cast<const float4*>(tab + S::T_E1 + n), e1 = *reinterpret_cast<const float4*>(tab + S::T_E1 + n + 4);
                const f32x4 &x = yacc[t][2 * bp], &z = yacc[t][2 * bp + 1];
                const float y[8] = {(x[0] - mean) * rstd * g0.x + e0.x, (x[1] - mean) * rstd * g0.y + e0.y, (x[2] - mean) * rstd * g0.z + e0.z, (x[3] - mean) * rstd * g0.w + e0.w,
                                    (z[0] - mean) * rstd * g1.x + e1.x, (z[1] - mean) * rstd * g1.y + e1.y, (z[2] - mean) * rstd * g1.z + e1.z, (z[3] - mean) * rstd * g1.w + e1.w};
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = ((float)oh[bp][i] + (float)ol[bp][i]) + y[i];
                uint4 o[2];
                split8<T, 2>(v, o);
                xh[t][bp] = as_v8<T>(o[0]);
                xl[t][bp] = as_v8<T>(o[1]);
            }
#pragma unroll
            for (int c = 0; c < CF; ++c) yacc[t][c] = f32x4{0.f, 0.f, 0.f, 0.f};
        }

        // ---- phase 2: the MLP ------------------------------------------------------------------------------------------------------------ //
        f32x4 hacc[FM][2];
        f32x2 gv[FM][4];                                // a chunk's pre-activations (bias added), then its activations
        uint4 hh[FM], hl[FM];
        auto zero_hacc = [&] {
#pragma unroll
            for (int t = 0; t < FM; ++t) { hacc[t][0] = f32x4{0.f, 0.f, 0.f, 0.f}; hacc[t][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
        };
        auto mm = [](v8 w, v8 x, f32x4 c) {               // probe 32: the operands are consumed, no matrix instruction
            if constexpr (P_MFMA) { c[0] += (float)w[0] + (float)x[0]; return c; }
            else return OpT<T>::mfma(w, x, c);
        };
        auto fc1_step = [&](int ks, const uint4& w0, const uint4& w1) {
#pragma unroll
            for (int t = 0; t < FM; ++t) hacc[t][0] = mm(as_v8<T>(w0), xh[t][ks], hacc[t][0]);
#pragma unroll
            for (int t = 0; t < FM; ++t) hacc[t][1] = mm(as_v8<T>(w1), xh[t][ks], hacc[t][1]);
            if constexpr (!S::ONE) {
#pragma unroll
                for (int t = 0; t < FM; ++t) hacc[t][0] = mm(as_v8<T>(w0), xl[t][ks], hacc[t][0]);
#pragma unroll
                for (int t = 0; t < FM; ++t) hacc[t][1] = mm(as_v8<T>(w1), xl[t][ks], hacc[t][1]);
            }
        };
        // hacc + bias of chunk j -> gv: the lane's 8 hidden units 16 n + 4 g + r of a row group, in fc2's k-slot order 8 g + 4 n + r
        auto take = [&](int j) {
            const float4 bb0 = *reinterpret_cast<const float4*>(tab + S::T_B1 + j * 32 + 4 * g), bb1 = *reinterpret_cast<const float4*>(tab + S::T_B1 + j * 32 + 16 + 4 * g);
#pragma unroll
            for (int t = 0; t < FM; ++t) {
                gv[t][0] = f32x2{hacc[t][0][0] + bb0.x, hacc[t][0][1] + bb0.y}; gv[t][1] = f32x2{hacc[t][0][2] + bb0.z, hacc[t][0][3] + bb0.w};
                gv[t][2] = f32x2{hacc[t][1][0] + bb1.x, hacc[t][1][1] + bb1.y}; gv[t][3] = f32x2{hacc[t][1][2] + bb1.z, hacc[t][1][3] + bb1.w};
            }
        };
        auto gelu_eval = [&](int e) { if constexpr (!P_GELU) gv[e >> 2][e & 3] = gelu_erf2(gv[e >> 2][e & 3]); };
        auto to_operand = [&](int t) {
            const float v[8] = {gv[t][0].x, gv[t][0].y, gv[t][1].x, gv[t][1].y, gv[t][2].x, gv[t][2].y, gv[t][3].x, gv[t][3].y};
            if constexpr (S::ONE) { uint4 o[1]; split8<T, 1>(v, o); hh[t] = o[0]; }
            else { uint4 o[2]; split8<T, 2>(v, o); hh[t] = o[0]; hl[t] = o[1]; }
        };
        auto fc2 = [&](const char* st) {
            sk_stream<CF / 2, RD, P_LDS>(st, [&](int p, const uint4& w0, const uint4& w1) {
#pragma unroll
                for (int t = 0; t < FM; ++t) yacc[t][2 * p] = mm(as_v8<T>(w0), as_v8<T>(hh[t]), yacc[t][2 * p]);
#pragma unroll
                for (int t = 0; t < FM; ++t) yacc[t][2 * p + 1] = mm(as_v8<T>(w1), as_v8<T>(hh[t]), yacc[t][2 * p + 1]);
                if constexpr (!S::ONE) {
#pragma unroll
                    for (int t = 0; t < FM; ++t) yacc[t][2 * p] = mm(as_v8<T>(w0), as_v8<T>(hl[t]), yacc[t][2 * p]);
#pragma unroll
                    for (int t = 0; t < FM; ++t) yacc[t][2 * p + 1] = mm(as_v8<T>(w1), as_v8<T>(hl[t]), yacc[t][2 * p + 1]);
                }
            });
        };

        top(NPU, false);                                // the residual loads above went through vector memory; this top requests a unit
        zero_hacc();
        sk_stream<KS, RD, P_LDS>(lrd + rp * S::UNIT, fc1_step);
        take(0);
        advance();
        pending = true;
        for (int j = 0; j < NCH; ++j) {
            pending = top(NPU + 1 + j, pending);
            const char* st = lrd + rp * S::UNIT;
            if constexpr (S::PIPE) {
                // fc1(j + 1) with the GELU of chunk j between its MFMAs (the last interval runs fc1 on the duplicate chunk the unit carries: discarded)
                zero_hacc();
                sk_stream<KS, RD, P_LDS>(st, [&](int ks, const uint4& w0, const uint4& w1) {
                    fc1_step(ks, w0, w1);
                    if (ks < S::NEK) {
#pragma unroll
                        for (int e = ks * S::EPK; e < (ks + 1) * S::EPK && e < S::NE; ++e) gelu_eval(e);
                    } else {
#pragma unroll
                        for (int t = (ks - S::NEK) * S::TPK; t < (ks - S::NEK + 1) * S::TPK && t < FM; ++t) to_operand(t);
                    }
                });
                fc2(st + S::SLOT);
                take(j + 1 < NCH ? j + 1 : j);
            } else {
#pragma unroll
                for (int e = 0; e < S::NE; ++e) gelu_eval(e);
#pragma unroll
                for (int t = 0; t < FM; ++t) to_operand(t);
                fc2(st + S::SLOT);
                if (j + 1 < NCH) {
                    zero_hacc();
                    sk_stream<KS, RD, P_LDS>(st, fc1_step);
                    take(j + 1);
                }
            }
            advance();
        }

        // ---- epilogue: + fc2 bias, LayerNorm(norm2), + x_mid (registers), whole blocks of the stream ------------------------------------ //
#pragma unroll
        for (int t = 0; t < FM; ++t) {
            float s = 0.f;
#pragma unroll
            for (int bp = 0; bp < KS; ++bp) {
                const int n = 32 * bp + 8 * g;
                const float4 b0 = *reinterpret_cast<const float4*>(tab + S::T_B2 + n), b1 = *reinterpret_cast<const float4*>(tab + S::T_B2 + n + 4);
                add8(yacc[t][2 * bp], yacc[t][2 * bp + 1], b0, b1);
#pragma unroll
                for (int r = 0; r < 4; ++r) s += yacc[t][2 * bp][r] + yacc[t][2 * bp + 1][r];
            }
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            const float mean = s * (1.0f / C);
            float q = 0.f;
#pragma unroll
            for (int c = 0; c < CF; ++c)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float d = yacc[t][c][r] - mean; q += d * d; }
            q += __shfl_xor(q, 16);
            q += __shfl_xor(q, 32);
            const float rstd = rsqrtf(q * (1.0f / C) + a.eps);
            if (!live[t] || (P_IO && rstd != 12345.f)) continue;
            T* dst = a.xs + ((rb0 + t) * KS << 9) + l15 * 32 + g * 8;
#pragma unroll
            for (int bp = 0; bp < KS; ++bp) {
                const int n = 32 * bp + 8 * g;
                const float4 g0 = *reinterpret_cast<const float4*>(tab + S::T_G2 + n), g1 = *reinterpret_cast<const float4*>(tab + S::T_G2 + n + 4);
                const float4 e0 = *reinterpret_cast<const float4*>(tab + S::T_E2 + n), e1 = *reinterpret_cast<const float4*>(tab + S::T_E2 + n + 4);
                float oh[8], ol[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) { oh[i] = (float)xh[t][bp][i]; ol[i] = (float)xl[t][bp][i]; }
                const f32x4 &x = yacc[t][2 * bp], &z = yacc[t][2 * bp + 1];
                const float v[8] = {(oh[0] + ol[0]) + ((x[0] - mean) * rstd * g0.x + e0.x), (oh[1] + ol[1]) + ((x[1] - mean) * rstd * g0.y + e0.y),
                                    (oh[2] + ol[2]) + ((x[2] - mean) * rstd * g0.z + e0.z), (oh[3] + ol[3]) + ((x[3] - mean) * rstd * g0.w + e0.w),
                                    (oh[4] + ol[4]) + ((z[0] - mean) * rstd * g1.x + e1.x), (oh[5] + ol[5]) + ((z[1] - mean) * rstd * g1.y + e1.y),
                                    (oh[6] + ol[6]) + ((z[2] - mean) * rstd * g1.z + e1.z), (oh[7] + ol[7]) + ((z[3] - mean) * rstd * g1.w + e1.w)};
                store8_planes<T, 2>(dst + (bp << 9), a.xs_plane, v);
            }
        }
    }
}

template <class T, class S>
static hipError_t launch_wide(const Block2Args<T>& a, hipStream_t s) {
    auto kern = proj_mlp_wide_kernel<T, S>;
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, n = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        e = hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, S::SMEM);
        if (e != hipSuccess) return e;
        cus = n;
    }
    const int ntiles = (a.M + S::BM - 1) / S::BM;
    if (ntiles == 0) return hipSuccess;
    const char* gv = getenv("SKP_WIDE_GRID");                   // tests: a small grid walks several tiles per workgroup on a small state
    const int want = gv ? atoi(gv) : cus;
    const unsigned grid = (unsigned)(ntiles < want ? ntiles : want);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(S::THREADS), S::SMEM, s, a, ntiles);
    return hipGetLastError();
}

// SKP_WIDE_PIPE=1 (measurement): the GELU spliced into fc1's k-steps
hipError_t op_proj_mlp_wide(const Geom& g, const BlockW<f16>& b, const int* winv, int res, f16* Xs, const Work<PrecF16x3>& wk, hipStream_t s, int one) {
    typedef f16 T;
    Block2Args<T> a{wk.ao, wk.ao_plane, g.ntok[res], Xs, wk.xs_plane[res], winv, b.projh, b.w1h, b.w2h,
                    b.proj_b, b.n1_g, b.n1_b, b.fc1_b, b.fc2_b, b.n2_g, b.n2_b, 1e-5f};
    if (a.M % 16 != 0) return hipErrorInvalidValue;
    const char* pv = getenv("SKP_WIDE_PIPE");
    const int pipe = pv ? atoi(pv) : 0;
    if (pipe) {
        if (res == 0) return one ? launch_wide<T, WideShape<192, 4, true, true>>(a, s) : launch_wide<T, WideShape<192, 4, false, true>>(a, s);
        return one ? launch_wide<T, WideShape<384, 2, true, true>>(a, s) : launch_wide<T, WideShape<384, 2, false, true>>(a, s);
    }
    if (res == 0) return one ? launch_wide<T, WideShape<192, 4, true, false>>(a, s) : launch_wide<T, WideShape<192, 4, false, false>>(a, s);
    return one ? launch_wide<T, WideShape<384, 2, true, false>>(a, s) : launch_wide<T, WideShape<384, 2, false, false>>(a, s);
}

}  // namespace skp
