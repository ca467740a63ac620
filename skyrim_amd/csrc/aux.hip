// One-time "prepare" kernels (fp32 master parameters -> kernel-ready operands) and the small
// row-statistics kernel of DownSample.  All HBM-bound, coalesced, no reuse worth staging.
#include "common.h"
#include "launchers.h"

namespace skp {

// dst[n][k] (ld = ldd, zero-filled for K <= k < ldd) = split(src[n*sn + k*sk]); generic strides cover
// plain [N][K] linears (sn=K, sk=1) and the ConvTranspose weights stored [K][N] (sn=1, sk=N).
template <class T, int NW>
__global__ void prep_weight_kernel(const float* __restrict__ src, T* __restrict__ dst, long long plane, int N, int K, int ldd,
                                   long long sn, long long sk, int blocked, int perm) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)N * ldd) return;
    const int n = (int)(i / ldd), k = (int)(i - (long long)n * ldd);
    const int ns = perm ? perm8_col(n) : n;       // prepared row n holds output column ns (common.h, perm8)
    const float v = k < K ? src[ns * sn + k * sk] : 0.f;
    const T h = (T)v;
    const long long o = blocked ? blk_off(n, k, ldd) : i;     // blocked: the DMA GEMMs' [N/16][K/32][16][32] layout
    dst[o] = h;
    if constexpr (NW == 2) dst[plane + o] = (T)(v - (float)h);
}

template <class T, int NW>
hipError_t prep_weight(const float* src, T* dst, long long plane, int N, int K, int ldd, long long sn, long long sk, int blocked, int perm, hipStream_t s) {
    if (perm && (N & 31)) return hipErrorInvalidValue;
    const long long total = (long long)N * ldd;
    hipLaunchKernelGGL((prep_weight_kernel<T, NW>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, dst, plane, N, K, ldd, sn, sk, blocked, perm);
    return hipGetLastError();
}
template hipError_t prep_weight<bf16, 2>(const float*, bf16*, long long, int, int, int, long long, long long, int, int, hipStream_t);
template hipError_t prep_weight<f16, 1>(const float*, f16*, long long, int, int, int, long long, long long, int, int, hipStream_t);
template hipError_t prep_weight<f16, 2>(const float*, f16*, long long, int, int, int, long long, long long, int, int, hipStream_t);

// Earth-specific bias gathered from the compact (3312, types, heads) table into the attention kernel's accumulator
// order (attention.hip, attn_key()): per (type, head, query fragment qf) one 2304-element tile laid out as
//   [kb = 0..3][lane][8]   keys 32 kb + 8 (lane >> 4) + [0..7]  (fragment pair 2kb, 2kb+1: one 16-byte load per lane)
//   [lane][4]              keys 128 + 4 (lane >> 4) + [0..3]    (fragment 8)
// with q = 16 qf + (lane & 15) and
//   index = (z_q + 2 z_k) * 23*36 + (h_q + 6 h_k) * 23 + (w_q - w_k + 11)   (pseudocode _construct_index)
// Odd (rolled) blocks fold the shifted-window mask in: mask_value where q and key sit in different Swin
// regions of the Z window / latitude window that mixes wrapped and unwrapped rows -- the LAST one when the block
// rolls by -(1,3,6) first (roll = -1), the FIRST one when it rolls by +(1,3,6) first (roll = +1); longitude is
// periodic -> never masked.
__global__ void prep_bias_expand_kernel(const float* __restrict__ table, f16* __restrict__ out, int types, int heads, int nH, int roll, float mask_value, int transposed) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)types * heads * 81 * 256;
    if (i >= total) return;
    const int e = (int)(i % 2304);
    long long rest = i / 2304;
    const int qf = (int)(rest % 9); rest /= 9;
    const int head = (int)(rest % heads);
    const int type = (int)(rest / heads);
    int lane, key;
    if (e < 2048) { const int kb = e >> 9; lane = (e >> 3) & 63; key = 32 * kb + 8 * (lane >> 4) + (e & 7); }
    else          { lane = ((e - 2048) >> 2) & 63; key = 128 + 4 * (lane >> 4) + (e & 3); }
    const int q = qf * 16 + (lane & 15);
    const int zq = q / 72, hq = (q / 12) % 6, wq = q % 12;
    const int zk = key / 72, hk = (key / 12) % 6, wk = key % 12;
    // transposed: the table is read as [key][query] -- the two meshgrid axes of position_index change roles
    const int idx = transposed ? (zk + 2 * zq) * (23 * 36) + (hk + 6 * hq) * 23 + (wk - wq + 11) : (zq + 2 * zk) * (23 * 36) + (hq + 6 * hk) * 23 + (wq - wk + 11);
    float v = table[((long long)idx * types + type) * heads + head];
    if (roll) {
        const int zi = type / nH, hi = type % nH;
        const int nZ = types / nH;
        const bool mz = (zi == (roll < 0 ? nZ - 1 : 0)) && (zq != zk);
        const bool mh = (hi == (roll < 0 ? nH - 1 : 0)) && ((hq < 3) != (hk < 3));
        if (mz || mh) v += mask_value;
    }
    out[i] = (f16)v;
}

hipError_t prep_bias_expand(const float* table, f16* out, int types, int heads, int nH, int roll, float mask_value, hipStream_t s, int transposed) {
    const long long total = (long long)types * heads * 81 * 256;
    hipLaunchKernelGGL(prep_bias_expand_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, table, out, types, heads, nH, roll, mask_value, transposed);
    return hipGetLastError();
}

// The same table in the COMPACT form the second attention kernel gathers from (attention.hip, earth_attention2_kernel): per (type, head)
// 144 rows r = (z_q + 2 z_k) 36 + (h_q + 6 h_k) of 24 fp16, column e = 22 - (w_q - w_k + 11) = w_k - w_q + 11 (reversed, so that
// four consecutive keys w_k .. w_k + 3 of one query are four consecutive ASCENDING entries), column 23 zero.  The shifted-window
// mask depends on (type, z_q, z_k, h_q, h_k) only -- exactly a row -- and is folded into the row.  6.9 KB per (type, head) where the
// expanded form takes 41 KB.
__global__ void prep_bias_compact_kernel(const float* __restrict__ table, f16* __restrict__ out, int types, int heads, int nH, int roll, float mask_value, int transposed) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)types * heads * 3456;
    if (i >= total) return;
    const int e = (int)(i % 24), r = (int)((i / 24) % 144);
    const long long th = i / 3456;
    const int head = (int)(th % heads), type = (int)(th / heads);
    if (e == 23) { out[i] = (f16)0.f; return; }
    const int rz = r / 36, rh = r % 36;
    const int zq = rz & 1, zk = rz >> 1, hq = rh % 6, hk = rh / 6;
    // row r, column e hold the bias of (z_q, z_k, h_q, h_k) and w_q - w_k + 11 = 22 - e; transposed: the table entry with the roles of
    // query and key exchanged, i.e. (z_k + 2 z_q, h_k + 6 h_q, w_k - w_q + 11 = e)
    const int src = transposed ? ((zk + 2 * zq) * 36 + (hk + 6 * hq)) * 23 + e : r * 23 + (22 - e);
    float v = table[((long long)src * types + type) * heads + head];
    if (roll) {
        const int zi = type / nH, hi = type % nH;
        const int nZ = types / nH;
        const bool mz = (zi == (roll < 0 ? nZ - 1 : 0)) && (zq != zk);
        const bool mh = (hi == (roll < 0 ? nH - 1 : 0)) && ((hq < 3) != (hk < 3));
        if (mz || mh) v += mask_value;
    }
    out[i] = (f16)v;
}

hipError_t prep_bias_compact(const float* table, f16* out, int types, int heads, int nH, int roll, float mask_value, hipStream_t s, int transposed) {
    const long long total = (long long)types * heads * 3456;
    hipLaunchKernelGGL(prep_bias_compact_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, table, out, types, heads, nH, roll, mask_value, transposed);
    return hipGetLastError();
}

// Window gather table: row m = win*144 + t of the (padded, rolled, window-partitioned) token grid
// -> source token row, or -1 for latitude padding.  win = (zi*nH + hi)*nW + wi, t = (tz*6 + th)*12 + tw.
__global__ void prep_window_index_kernel(int* __restrict__ idx, int Z, int H, int W, int Hp, int top, int roll, int surface_last) {
    const int nH = Hp / 6, nW = W / 12;
    const long long total = (long long)(Z / 2) * nH * nW * WIN_TOKENS;
    const long long m = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= total) return;
    const int t = (int)(m % WIN_TOKENS);
    const int win = (int)(m / WIN_TOKENS);
    const int wi = win % nW, hi = (win / nW) % nH, zi = win / (nW * nH);
    int pz = 2 * zi + t / 72, ph = 6 * hi + (t / 12) % 6, pw = 12 * wi + t % 12;
    // rolled[p] = x[(p - roll * shift) mod n]  (torch.roll semantics; roll = -1 | +1 selects the direction of the first roll)
    if (roll < 0) { pz = (pz + 1) % Z; ph = (ph + 3) % Hp; pw = (pw + 6) % W; }
    if (roll > 0) { pz = (pz + Z - 1) % Z; ph = (ph + Hp - 3) % Hp; pw = (pw + W - 6) % W; }
    const int h = ph - top;
    // pz is the LOGICAL level (windows, roll and mask act on it); the stream stores the surface slab at level 0 in either convention:
    // with the surface as the last logical level, logical k < Z - 1 is upper-air slab k = storage k + 1, logical Z - 1 is storage 0
    const int sz = surface_last ? (pz + 1) % Z : pz;
    idx[m] = (h >= 0 && h < H) ? (sz * H + h) * W + pw : -1;
}

hipError_t prep_window_index(int* idx, int Z, int H, int W, int Hp, int top, int roll, hipStream_t s, int surface_last) {
    const long long total = (long long)(Z / 2) * (Hp / 6) * (W / 12) * WIN_TOKENS;
    hipLaunchKernelGGL(prep_window_index_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, idx, Z, H, W, Hp, top, roll, surface_last);
    return hipGetLastError();
}

// qkv Linear stored with its 3C output rows packed (heads, 3, head_dim): rewritten in the (3, heads, head_dim) order every kernel assumes
__global__ void prep_qkv_rows_kernel(const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ wo, float* __restrict__ bo, int C, int heads) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = 3LL * C * C;
    if (i >= total) return;
    const int k = (int)(i % C), row = (int)(i / C);                // canonical row = which * C + head * hd + d
    const int hd = C / heads, which = row / C, head = (row % C) / hd, d = row % hd;
    const int src = (head * 3 + which) * hd + d;
    wo[i] = w[(long long)src * C + k];
    if (k == 0) bo[row] = b[src];
}
hipError_t prep_qkv_rows(const float* w, const float* b, float* w_out, float* b_out, int C, int heads, hipStream_t s) {
    const long long total = 3LL * C * C;
    hipLaunchKernelGGL(prep_qkv_rows_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, b, w_out, b_out, C, heads);
    return hipGetLastError();
}

// inverse of the window table: every stream token sits in exactly one (non-padding) window row
__global__ void prep_window_inverse_kernel(const int* __restrict__ idx, int n, int* __restrict__ inv) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m < n && idx[m] >= 0) inv[idx[m]] = m;
}
hipError_t prep_window_inverse(const int* idx, int n, int* inv, hipStream_t s) {
    hipLaunchKernelGGL(prep_window_inverse_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, idx, n, inv);
    return hipGetLastError();
}

__global__ void prep_reciprocal_kernel(const float* __restrict__ src, float* __restrict__ dst, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = 1.0f / src[i];
}
hipError_t prep_reciprocal(const float* src, float* dst, int n, hipStream_t s) {
    hipLaunchKernelGGL(prep_reciprocal_kernel, dim3((n + 255) / 256), dim3(256), 0, s, src, dst, n);
    return hipGetLastError();
}

// ---- calibration of a two-term plan (api.hip Engine::calibrate) ---- //
// column sums of one GEMM operand in the blocked plane layout (hi, or hi + lo), over `rows` rows taken through `rowmap` when given:
// per-workgroup partial sums, then one thread per column adds them in workgroup order (no atomics: the same state gives the same bits)
template <class T>
__global__ void colsum_planes_kernel(const T* __restrict__ x, long long plane, int nplanes, const int* __restrict__ rowmap, int rows, int K,
                                     int rows_per_wg, float* __restrict__ partial) {
    const int r0 = blockIdx.x * rows_per_wg, r1 = min(rows, r0 + rows_per_wg);
    for (int c = threadIdx.x; c < K; c += blockDim.x) {
        float acc = 0.f;
        for (int r = r0; r < r1; ++r) {
            const long long row = rowmap ? rowmap[r] : r;
            const T* p = x + blk_off(row, c, K);
            acc += (float)p[0];
            if (nplanes == 2) acc += (float)p[plane];
        }
        partial[(long long)blockIdx.x * K + c] = acc;
    }
}
__global__ void colsum_reduce_kernel(const float* __restrict__ partial, int nwg, int K, float* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= K) return;
    float acc = 0.f;
    for (int i = 0; i < nwg; ++i) acc += partial[(long long)i * K + c];
    out[c] = acc;
}
size_t colsum_scratch_floats(int rows, int K) { return (size_t)((rows + 127) / 128) * K; }
template <class T>
hipError_t colsum_planes(const T* x, long long plane, int nplanes, const int* rowmap, int rows, int K, float* scratch, float* out, hipStream_t s) {
    const int per = 128, nwg = (rows + per - 1) / per;
    hipLaunchKernelGGL((colsum_planes_kernel<T>), dim3((unsigned)nwg), dim3(256), 0, s, x, plane, nplanes, rowmap, rows, K, per, scratch);
    hipLaunchKernelGGL(colsum_reduce_kernel, dim3((K + 63) / 64), dim3(64), 0, s, scratch, nwg, K, out);
    return hipGetLastError();
}
template hipError_t colsum_planes<f16>(const f16*, long long, int, const int*, int, int, float*, float*, hipStream_t);
template hipError_t colsum_planes<bf16>(const bf16*, long long, int, const int*, int, int, float*, float*, hipStream_t);

// bias[n] += sum_k (w[n][k] - fp16(w[n][k])) * colsum[k] * scale: the mean over the calibration rows of the term a one-plane weight drops
__global__ void bias_fold_kernel(const float* __restrict__ w, const float* __restrict__ colsum, float scale, float* __restrict__ bias, int N, int K) {
    const int n = blockIdx.x, lane = threadIdx.x;
    float acc = 0.f;
    for (int k = lane; k < K; k += 64) {
        const float v = w[(long long)n * K + k];
        acc += (v - (float)(f16)v) * colsum[k];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) bias[n] += acc * scale;
}
hipError_t bias_fold(const float* w, const float* colsum, float scale, float* bias, int N, int K, hipStream_t s) {
    hipLaunchKernelGGL(bias_fold_kernel, dim3(N), dim3(64), 0, s, w, colsum, scale, bias, N, K);
    return hipGetLastError();
}

// fp32 -> 16-bit hi/lo planes (shadow of a residual stream handed in through the stage-level API)
template <class T, int NPL>
__global__ void split_planes_kernel(const float* __restrict__ x, T* __restrict__ planes, long long plane, long long n4, int C) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    const float vv[4] = {v.x, v.y, v.z, v.w};
    const long long e = i * 4, row = e / C;
    store4_planes<T, NPL>(planes + blk_off(row, (int)(e - row * C), C), plane, vv);
}
template <class T, int NPL>
hipError_t split_planes(const float* x, T* planes, long long plane, long long n, int C, hipStream_t s) {
    const long long n4 = n / 4;
    hipLaunchKernelGGL((split_planes_kernel<T, NPL>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, x, planes, plane, n4, C);
    return hipGetLastError();
}
template hipError_t split_planes<bf16, 2>(const float*, bf16*, long long, long long, int, hipStream_t);
template hipError_t split_planes<f16, 1>(const float*, f16*, long long, long long, int, hipStream_t);
template hipError_t split_planes<f16, 2>(const float*, f16*, long long, long long, int, hipStream_t);

// planes -> fp32 row-major (stage-level API / tests only: the step itself never materialises an fp32 stream)
template <class T>
__global__ void merge_planes_kernel(const T* __restrict__ planes, long long plane, float* __restrict__ x, long long n4, int C) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    typedef T t4 __attribute__((ext_vector_type(4)));
    const long long e = i * 4, row = e / C;
    const T* p = planes + blk_off(row, (int)(e - row * C), C);
    const t4 h = __builtin_bit_cast(t4, *reinterpret_cast<const uint2*>(p));
    const t4 l = __builtin_bit_cast(t4, *reinterpret_cast<const uint2*>(p + plane));
    reinterpret_cast<float4*>(x)[i] = make_float4((float)h[0] + (float)l[0], (float)h[1] + (float)l[1], (float)h[2] + (float)l[2], (float)h[3] + (float)l[3]);
}
template <class T>
hipError_t merge_planes(const T* planes, long long plane, float* x, long long n, int C, hipStream_t s) {
    const long long n4 = n / 4;
    hipLaunchKernelGGL((merge_planes_kernel<T>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, planes, plane, x, n4, C);
    return hipGetLastError();
}
template hipError_t merge_planes<bf16>(const bf16*, long long, float*, long long, int, hipStream_t);
template hipError_t merge_planes<f16>(const f16*, long long, float*, long long, int, hipStream_t);

// DownSample LayerNorm(4C) statistics of the 2x2-merged rows: one wavefront per merged row (z, h', w'); the 4 source
// tokens are read from the residual planes (hi + lo) in 8-element chunks (the one below the grid is zero padding).
template <class T>
__global__ void __launch_bounds__(256) merge_stats_kernel(const T* __restrict__ x, long long plane, float2* __restrict__ stats, int Z, int H1,
                                                           int W1, int H2, int W2, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const long long m = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long M = (long long)Z * H2 * W2;
    if (m >= M) return;
    const int hw = H2 * W2;
    const int z = (int)(m / hw), rem = (int)(m - (long long)z * hw);
    const int h = rem / W2, w = rem - h * W2;
    const int c8 = C / 8;                      // 16-byte chunks per token
    const int nchunks = 4 * c8;                // per merged row (<= 384 -> up to 6 chunks per lane)
    float vals[6][8];
    float s = 0.f;
#pragma unroll
    for (int it = 0; it < 6; ++it) {
        const int j = it * 64 + lane;
#pragma unroll
        for (int e = 0; e < 8; ++e) vals[it][e] = 0.f;
        if (j < nchunks) {
            const int qd = j / c8, c = (j - qd * c8) * 8;
            const int hf = 2 * h + (qd >> 1), wf = 2 * w + (qd & 1);
            if (hf < H1) {
                const T* p = x + blk_off(((long long)z * H1 + hf) * W1 + wf, c, C);
                const typename OpT<T>::v8 hi = as_v8<T>(*reinterpret_cast<const uint4*>(p));
                const typename OpT<T>::v8 lo = as_v8<T>(*reinterpret_cast<const uint4*>(p + plane));
#pragma unroll
                for (int e = 0; e < 8; ++e) { vals[it][e] = (float)hi[e] + (float)lo[e]; s += vals[it][e]; }
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (4.0f * C);
    float ss = 0.f;
#pragma unroll
    for (int it = 0; it < 6; ++it) {
        if (it * 64 + lane < nchunks) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = vals[it][e] - mean; ss += d * d; }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
    if (lane == 0) stats[m] = make_float2(mean, rsqrtf(ss / (4.0f * C) + eps));
}

template <class T>
hipError_t merge_stats(const T* x, long long plane, float2* stats, int Z, int H1, int W1, int H2, int W2, int C, float eps, hipStream_t s) {
    const long long M = (long long)Z * H2 * W2;
    hipLaunchKernelGGL((merge_stats_kernel<T>), dim3((unsigned)((M + 3) / 4)), dim3(256), 0, s, x, plane, stats, Z, H1, W1, H2, W2, C, eps);
    return hipGetLastError();
}
template hipError_t merge_stats<bf16>(const bf16*, long long, float2*, int, int, int, int, int, int, float, hipStream_t);
template hipError_t merge_stats<f16>(const f16*, long long, float2*, int, int, int, int, int, int, float, hipStream_t);

}  // namespace skp
