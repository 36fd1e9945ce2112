// xq_disc.hip — fused kernels of the DinoDisc discriminator heads (gfx950), token-major activations [B][L][C].
//
// Replaces, per head block of the reference (tokenizer/tokenizer_image/discriminator_dino.py:127-154 BatchNormLocal,
// :157-166 make_block = SpectralConv1d -> BatchNormLocal -> LeakyReLU(0.2), :113-119 ResidualBlock), the ~12 unfused fp32
// ATen passes per block forward (float(), mean, var, sub, add, sqrt, div, mul, add, leaky_relu, add, mul) and their ~30
// autograd passes backward, plus F.pad(circular) + unfold / the generic unfold backward of the kernel-9 conv:
//   bnlocal_lrelu_fwd : per virtual batch g (8 samples) and channel c: mean/var over the 8*L tokens (fp32, two-pass),
//                       out = [ (lrelu((y - mean) * rstd * w + b) + skip) * ratio ]   (skip/ratio: the ResidualBlock)
//   bnlocal_lrelu_bwd : the transpose: g_y, g_skip, per-group partial sums of g_w / g_b
//   unfold1d_circular : cols[b, l, tap, :] = h[b, (l + tap - K/2) mod L, :]   (the im2col of the circular conv, so that the
//                       conv itself is one library GEMM with the taps in the reduction)
//   fold1d_circular   : dh[b, l, :] = sum_tap dcols[b, (l - tap + K/2) mod L, tap, :]
// One block owns (group g) x (CH channels), CH = 64, 32 or 16 — the widest that still gives >= 256 blocks (round 4: at C = 384 and 16
// virtual batches the fixed 64-channel blocks made a 96-block grid on 256 CUs, each walking its 1568 rows three times: 55 - 75 us per
// call for 40 - 60 MB); its 8*L x CH slab stays in L2 across the passes.
#include <stdlib.h>
#include "xq_common.hpp"
#include "xq_internal.hpp"
#include "../../include/xq_ops.h"

#include "xq_vec.hpp"

using namespace xq;

static constexpr int BN_CH = 64;   // widest channel block (and the granularity C must have)
static constexpr int BN_UNR = 4;   // rows fetched per trip of a row walk before any is consumed

// column sums over the block's rows: thread (rt, ct) owns VEC columns, rows rt, rt+RPP, ...; reduce over rt through LDS
template <int VEC, int TPR, int RPP, int CH>
__device__ __forceinline__ void block_colsum(float (&acc)[VEC], float *lds /* [RPP][CH] */, float *out /* [CH] in LDS */) {
    const int ct = threadIdx.x % TPR, rt = threadIdx.x / TPR;
#pragma unroll
    for (int j = 0; j < VEC; ++j) lds[rt * CH + ct * VEC + j] = acc[j];
    __syncthreads();
    if (RPP * TPR > 256) {
        // wide blocks (round 5): halving tree over the row groups, every thread on its own VEC columns — fixed order, log2(RPP) short steps
        for (int s = RPP / 2; s > 0; s >>= 1) {
            if (rt < s) {
#pragma unroll
                for (int j = 0; j < VEC; ++j) lds[rt * CH + ct * VEC + j] += lds[(rt + s) * CH + ct * VEC + j];
            }
            __syncthreads();
        }
        if (threadIdx.x < CH) out[threadIdx.x] = lds[threadIdx.x];
    } else if (threadIdx.x < CH) {
        float s = 0.0f;
        for (int r = 0; r < RPP; ++r) s += lds[r * CH + threadIdx.x];
        out[threadIdx.x] = s;
    }
    __syncthreads();
}

template <typename T, int CH, int NT = 256>
__global__ __launch_bounds__(NT) void bnlocal_lrelu_fwd_kernel(const T *__restrict__ y, const float *__restrict__ w, const float *__restrict__ b,
                                                                const T *__restrict__ skip, int R, int C, float eps, float slope, float ratio,
                                                                T *__restrict__ out, float *__restrict__ mean_out, float *__restrict__ rstd_out) {
    constexpr int VEC = 16 / sizeof(T), TPR = CH / VEC, RPP = NT / TPR;
    __shared__ float red[RPP * CH];
    __shared__ float stat[2][CH];
    const int g = blockIdx.y, c0 = blockIdx.x * CH;
    const int ct = threadIdx.x % TPR, rt = threadIdx.x / TPR;
    const T *yb = y + (long)g * R * C + c0 + ct * VEC;
    float acc[VEC], v[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = 0.0f;
    // Round 5: the row walks fetch BN_UNR rows per trip before consuming any (a block has 256 threads = 128 rows in flight at CH = 16 and
    // walks 1568 rows: with one load per trip every pass was a chain of 12 dependent L2 / HBM round trips — 47 us for 38 MB; the rows of a
    // trip are summed in the same ascending order as before, so the statistics are bit-identical)
    for (int r = rt; r < R; r += BN_UNR * RPP) {
        float vv[BN_UNR][VEC];
#pragma unroll
        for (int u = 0; u < BN_UNR; ++u) {
            const int ru = r + u * RPP;
            load_vec<T, VEC>(yb + (long)(ru < R ? ru : r) * C, vv[u]);
        }
#pragma unroll
        for (int u = 0; u < BN_UNR; ++u)
            if (r + u * RPP < R) {
#pragma unroll
                for (int j = 0; j < VEC; ++j) acc[j] += vv[u][j];
            }
    }
    block_colsum<VEC, TPR, RPP, CH>(acc, red, stat[0]);
    float mu[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { mu[j] = stat[0][ct * VEC + j] / (float)R; acc[j] = 0.0f; }
    for (int r = rt; r < R; r += BN_UNR * RPP) {
        float vv[BN_UNR][VEC];
#pragma unroll
        for (int u = 0; u < BN_UNR; ++u) {
            const int ru = r + u * RPP;
            load_vec<T, VEC>(yb + (long)(ru < R ? ru : r) * C, vv[u]);
        }
#pragma unroll
        for (int u = 0; u < BN_UNR; ++u)
            if (r + u * RPP < R) {
#pragma unroll
                for (int j = 0; j < VEC; ++j) { const float d = vv[u][j] - mu[j]; acc[j] = fmaf(d, d, acc[j]); }
            }
    }
    block_colsum<VEC, TPR, RPP, CH>(acc, red, stat[1]);
    float rs[VEC], ww[VEC], bb[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        rs[j] = 1.0f / sqrtf(stat[1][ct * VEC + j] / (float)R + eps);
        ww[j] = w ? w[c0 + ct * VEC + j] : 1.0f;
        bb[j] = b ? b[c0 + ct * VEC + j] : 0.0f;
    }
    if (rt == 0) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            mean_out[(long)g * C + c0 + ct * VEC + j] = mu[j];
            rstd_out[(long)g * C + c0 + ct * VEC + j] = rs[j];
        }
    }
    T *ob = out + (long)g * R * C + c0 + ct * VEC;
    const T *sb = skip ? skip + (long)g * R * C + c0 + ct * VEC : nullptr;
    for (int r = rt; r < R; r += BN_UNR * RPP) {
        float vv[BN_UNR][VEC], sk[BN_UNR][VEC];
#pragma unroll
        for (int u = 0; u < BN_UNR; ++u) {
            const int ru = r + u * RPP < R ? r + u * RPP : r;
            load_vec<T, VEC>(yb + (long)ru * C, vv[u]);
            if (sb) load_vec<T, VEC>(sb + (long)ru * C, sk[u]);
        }
#pragma unroll
        for (int u = 0; u < BN_UNR; ++u)
            if (r + u * RPP < R) {
                float o[VEC];
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const float pre = fmaf((vv[u][j] - mu[j]) * rs[j], ww[j], bb[j]);
                    const float a = pre > 0.0f ? pre : pre * slope;
                    o[j] = sb ? (a + sk[u][j]) * ratio : a;
                }
                store_vec<T, VEC>(ob + (long)(r + u * RPP) * C, o);
            }
    }
    (void)v;
}

template <typename T, int CH, int NT = 256>
__global__ __launch_bounds__(NT) void bnlocal_lrelu_bwd_kernel(const T *__restrict__ g_out, const T *__restrict__ y, const float *__restrict__ w,
                                                                const float *__restrict__ b, const float *__restrict__ mean,
                                                                const float *__restrict__ rstd, int R, int C, float slope, float ratio,
                                                                int has_skip, T *__restrict__ g_y, T *__restrict__ g_skip,
                                                                float *__restrict__ gw_part, float *__restrict__ gb_part) {
    constexpr int VEC = 16 / sizeof(T), TPR = CH / VEC, RPP = NT / TPR;
    __shared__ float red[RPP * CH];
    __shared__ float stat[2][CH];
    const int g = blockIdx.y, c0 = blockIdx.x * CH;
    const int ct = threadIdx.x % TPR, rt = threadIdx.x / TPR;
    const long off = (long)g * R * C + c0 + ct * VEC;
    const float scale = has_skip ? ratio : 1.0f;
    float mu[VEC], rs[VEC], ww[VEC], bb[VEC], s1[VEC], s2[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const int c = c0 + ct * VEC + j;
        mu[j] = mean[(long)g * C + c];
        rs[j] = rstd[(long)g * C + c];
        ww[j] = w ? w[c] : 1.0f;
        bb[j] = b ? b[c] : 0.0f;
        s1[j] = 0.0f;
        s2[j] = 0.0f;
    }
    for (int r = rt; r < R; r += BN_UNR * RPP) {
        float v[BN_UNR][VEC], go[BN_UNR][VEC];
#pragma unroll
        for (int u = 0; u < BN_UNR; ++u) {
            const int ru = r + u * RPP < R ? r + u * RPP : r;
            load_vec<T, VEC>(y + off + (long)ru * C, v[u]);
            load_vec<T, VEC>(g_out + off + (long)ru * C, go[u]);
        }
#pragma unroll
        for (int u = 0; u < BN_UNR; ++u)
            if (r + u * RPP < R) {
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const float z = (v[u][j] - mu[j]) * rs[j];
                    const float pre = fmaf(z, ww[j], bb[j]);
                    const float gp = go[u][j] * scale * (pre > 0.0f ? 1.0f : slope);
                    s1[j] += gp;
                    s2[j] = fmaf(gp, z, s2[j]);
                }
            }
    }
    block_colsum<VEC, TPR, RPP, CH>(s1, red, stat[0]);
    block_colsum<VEC, TPR, RPP, CH>(s2, red, stat[1]);
    float m1[VEC], m2[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        m1[j] = stat[0][ct * VEC + j] / (float)R;
        m2[j] = stat[1][ct * VEC + j] / (float)R;
    }
    if (rt == 0) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            gb_part[(long)g * C + c0 + ct * VEC + j] = stat[0][ct * VEC + j];
            gw_part[(long)g * C + c0 + ct * VEC + j] = stat[1][ct * VEC + j];
        }
    }
    for (int r = rt; r < R; r += BN_UNR * RPP) {
        float v[BN_UNR][VEC], go[BN_UNR][VEC];
#pragma unroll
        for (int u = 0; u < BN_UNR; ++u) {
            const int ru = r + u * RPP < R ? r + u * RPP : r;
            load_vec<T, VEC>(y + off + (long)ru * C, v[u]);
            load_vec<T, VEC>(g_out + off + (long)ru * C, go[u]);
        }
#pragma unroll
        for (int u = 0; u < BN_UNR; ++u)
            if (r + u * RPP < R) {
                float gy[VEC], gs[VEC];
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const float z = (v[u][j] - mu[j]) * rs[j];
                    const float pre = fmaf(z, ww[j], bb[j]);
                    const float gp = go[u][j] * scale * (pre > 0.0f ? 1.0f : slope);
                    gy[j] = rs[j] * ww[j] * (gp - m1[j] - z * m2[j]);
                    gs[j] = go[u][j] * scale;
                }
                store_vec<T, VEC>(g_y + off + (long)(r + u * RPP) * C, gy);
                if (has_skip) store_vec<T, VEC>(g_skip + off + (long)(r + u * RPP) * C, gs);
            }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void unfold1d_kernel(const T *__restrict__ h, int B, int L, int C, int K, T *__restrict__ cols) {
    constexpr int VEC = 16 / sizeof(T);
    const int cv = C / VEC;
    const long total = (long)B * L * K * cv;
    const int pad = K / 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % cv);
        long t = i / cv;
        const int tap = (int)(t % K);
        t /= K;
        const int l = (int)(t % L);
        const long bb = t / L;
        int ls = l + tap - pad;
        ls = ls < 0 ? ls + L : (ls >= L ? ls - L : ls);
        const Pack<T, VEC> v = *reinterpret_cast<const Pack<T, VEC> *>(h + ((bb * L + ls) * C + (long)c * VEC));
        *reinterpret_cast<Pack<T, VEC> *>(cols + i * VEC) = v;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void fold1d_kernel(const T *__restrict__ dcols, int B, int L, int C, int K, T *__restrict__ dh) {
    constexpr int VEC = 16 / sizeof(T);
    const int cv = C / VEC;
    const long total = (long)B * L * cv;
    const int pad = K / 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % cv);
        long t = i / cv;
        const int l = (int)(t % L);
        const long bb = t / L;
        float acc[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] = 0.0f;
        for (int tap = 0; tap < K; ++tap) {
            int ls = l - tap + pad;
            ls = ls < 0 ? ls + L : (ls >= L ? ls - L : ls);
            float v[VEC];
            load_vec<T, VEC>(dcols + (((bb * L + ls) * K + tap) * C + (long)c * VEC), v);
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc[j] += v[j];
        }
        store_vec<T, VEC>(dh + i * VEC, acc);
    }
}

// channels per block: the widest of 64 / 32 / 16 that fills the chip (C % 64 == 0 by contract)
static int bn_block_channels(int C, int G) {
    for (int ch = 64; ch > 16; ch >>= 1)
        if ((long)(C / ch) * G >= num_cus()) return ch;
    return 16;
}

// XQ_BN_WIDE=0: the round-4 grid (16-channel blocks of 256 threads) for A/B timing
static bool bn_wide() {
    static const int v = getenv("XQ_BN_WIDE") ? atoi(getenv("XQ_BN_WIDE")) : 1;
    return v != 0;
}

static int bn_check(const char *fn, int G, int R, int C) {
    if (G < 0 || R < 1 || C < 1 || C % BN_CH != 0)
        return xq_set_error(XQ_EINVAL, "%s: needs rows_per_group >= 1 and C %% 64 == 0 (got %ld, %ld)", fn, (long)R, (long)C);
    return XQ_OK;
}

extern "C" int xq_bnlocal_lrelu_forward(const void *y, const float *w, const float *b, const void *skip, int G, int rows_per_group, int C,
                                        int act_bf16, float eps, float slope, float ratio, void *out, float *mean, float *rstd,
                                        xq_stream_t stream) {
    const char *fn = "xq_bnlocal_lrelu_forward";
    if (int rc = bn_check(fn, G, rows_per_group, C)) return rc;
    if (G == 0) return XQ_OK;
    if (!y || !out || !mean || !rstd) return xq_set_error(XQ_EINVAL, "%s: null pointer", fn);
    hipStream_t s = (hipStream_t)stream;
    const int ch = bn_block_channels(C, G);
    if (act_bf16 && ch < 64 && bn_wide()) {
        // round 5: 64-channel blocks of 1024 threads — the same number of waves as the 16-channel grid, but every row access is a full
        // 128-byte line (16-channel blocks read 32 bytes of each 768-byte row: half of every fetched line was wasted)
        hipLaunchKernelGGL((bnlocal_lrelu_fwd_kernel<bf16, 64, 1024>), dim3(C / 64, G), dim3(1024), 0, s, (const bf16 *)y, w, b, (const bf16 *)skip,
                           rows_per_group, C, eps, slope, ratio, (bf16 *)out, mean, rstd);
        return xq_check_launch(fn);
    }
#define BN_FWD(T_, CH_)                                                                                                                          \
    hipLaunchKernelGGL((bnlocal_lrelu_fwd_kernel<T_, CH_>), dim3(C / CH_, G), dim3(256), 0, s, (const T_ *)y, w, b, (const T_ *)skip, rows_per_group, \
                       C, eps, slope, ratio, (T_ *)out, mean, rstd)
    if (act_bf16) { if (ch == 64) BN_FWD(bf16, 64); else if (ch == 32) BN_FWD(bf16, 32); else BN_FWD(bf16, 16); }
    else { if (ch == 64) BN_FWD(float, 64); else if (ch == 32) BN_FWD(float, 32); else BN_FWD(float, 16); }
#undef BN_FWD
    return xq_check_launch(fn);
}

extern "C" int xq_bnlocal_lrelu_backward(const void *g_out, const void *y, const float *w, const float *b, const float *mean, const float *rstd,
                                         int G, int rows_per_group, int C, int act_bf16, float slope, float ratio, int has_skip, void *g_y,
                                         void *g_skip, float *gw_part, float *gb_part, xq_stream_t stream) {
    const char *fn = "xq_bnlocal_lrelu_backward";
    if (int rc = bn_check(fn, G, rows_per_group, C)) return rc;
    if (G == 0) return XQ_OK;
    if (!g_out || !y || !mean || !rstd || !g_y || !gw_part || !gb_part || (has_skip && !g_skip))
        return xq_set_error(XQ_EINVAL, "%s: null pointer", fn);
    hipStream_t s = (hipStream_t)stream;
    const int ch = bn_block_channels(C, G);
    if (act_bf16 && ch < 64 && bn_wide()) {
        hipLaunchKernelGGL((bnlocal_lrelu_bwd_kernel<bf16, 64, 1024>), dim3(C / 64, G), dim3(1024), 0, s, (const bf16 *)g_out, (const bf16 *)y, w, b, mean,
                           rstd, rows_per_group, C, slope, ratio, has_skip, (bf16 *)g_y, (bf16 *)g_skip, gw_part, gb_part);
        return xq_check_launch(fn);
    }
#define BN_BWD(T_, CH_)                                                                                                                          \
    hipLaunchKernelGGL((bnlocal_lrelu_bwd_kernel<T_, CH_>), dim3(C / CH_, G), dim3(256), 0, s, (const T_ *)g_out, (const T_ *)y, w, b, mean, rstd,   \
                       rows_per_group, C, slope, ratio, has_skip, (T_ *)g_y, (T_ *)g_skip, gw_part, gb_part)
    if (act_bf16) { if (ch == 64) BN_BWD(bf16, 64); else if (ch == 32) BN_BWD(bf16, 32); else BN_BWD(bf16, 16); }
    else { if (ch == 64) BN_BWD(float, 64); else if (ch == 32) BN_BWD(float, 32); else BN_BWD(float, 16); }
#undef BN_BWD
    return xq_check_launch(fn);
}

static int fold_check(const char *fn, int B, int L, int C, int K, int act_bf16) {
    const int vec = act_bf16 ? 8 : 4;
    if (B < 0 || L < 1 || K < 1 || K > L || C < 1 || C % vec != 0)
        return xq_set_error(XQ_EINVAL, "%s: needs 1 <= K <= L and C a multiple of the 16-byte vector (K=%ld, C=%ld)", fn, (long)K, (long)C);
    return XQ_OK;
}

// ---- class-token readout of the frozen DINO trunk (discriminator_dino.py:339-347: acts.append((x[:, 1:] + x[:, :1]) ...)) -------------------
// out[b][l][:] = t[b][l + 1][:] + t[b][0][:] from the fp32 residual stream t [B][L + 1][C], written in the heads' activation dtype (one pass
// instead of an fp32 add + a cast); backward: gt[b][l + 1][:] = g[b][l][:], gt[b][0][:] = sum_l g[b][l][:] in a fixed order.
template <typename T>
__global__ __launch_bounds__(256) void cls_readout_fwd_kernel(const float *__restrict__ t, int B, int L, int C, T *__restrict__ out) {
    constexpr int VEC = 16 / sizeof(T);
    const int cv = C / VEC;
    const long total = (long)B * L * cv;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % cv) * VEC;
        const long row = i / cv;
        const long b = row / L, l = row - b * L;
        const float *cls = t + b * (long)(L + 1) * C + c, *tok = cls + (l + 1) * (long)C;
        float v[VEC];
#pragma unroll
        for (int j = 0; j < VEC; j += 4) {
            const float4 a = *reinterpret_cast<const float4 *>(tok + j), k = *reinterpret_cast<const float4 *>(cls + j);
            v[j] = a.x + k.x; v[j + 1] = a.y + k.y; v[j + 2] = a.z + k.z; v[j + 3] = a.w + k.w;
        }
        store_vec<T, VEC>(out + row * C + c, v);
    }
}

// block (b, 64-column chunk): 8 column octets x 32 row lanes; each lane walks the tokens l = lane, lane + 32, ...
template <typename T>
__global__ __launch_bounds__(256) void cls_readout_bwd_kernel(const T *__restrict__ g, int L, int C, float *__restrict__ gt) {
    const int b = blockIdx.x, c = blockIdx.y * 64 + (threadIdx.x & 7) * 8, rl = threadIdx.x >> 3;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bool live = c < C;      // C % 8 == 0
    if (live) {
        for (int l = rl; l < L; l += 32) {
            float v[8];
            if (sizeof(T) == 2) load_vec<T, 8>(g + ((long)b * L + l) * C + c, v);
            else { load_vec<T, 4>(g + ((long)b * L + l) * C + c, reinterpret_cast<float(&)[4]>(v[0])); load_vec<T, 4>(g + ((long)b * L + l) * C + c + 4, reinterpret_cast<float(&)[4]>(v[4])); }
            float *o = gt + ((long)b * (L + 1) + l + 1) * C + c;
            *reinterpret_cast<float4 *>(o) = make_float4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<float4 *>(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += v[j];
        }
    }
    __shared__ float red[32][65];
#pragma unroll
    for (int j = 0; j < 8; ++j) red[rl][(threadIdx.x & 7) * 8 + j] = acc[j];
    __syncthreads();
    if (threadIdx.x < 64 && blockIdx.y * 64 + threadIdx.x < C) {
        float s = 0.0f;
#pragma unroll
        for (int r = 0; r < 32; ++r) s += red[r][threadIdx.x];
        gt[(long)b * (L + 1) * C + blockIdx.y * 64 + threadIdx.x] = s;
    }
}

extern "C" int xq_cls_readout_forward(const float *t, int B, int L, int C, int act_bf16, void *out, xq_stream_t stream) {
    const char *fn = "xq_cls_readout_forward";
    if (B < 0 || L < 1 || C < 8 || C % 8) return xq_set_error(XQ_EINVAL, "%s: bad shape (C %% 8 == 0)", fn);
    if (B == 0) return XQ_OK;
    if (!t || !out) return xq_set_error(XQ_EINVAL, "%s: null pointer", fn);
    const long total = (long)B * L * (C / (act_bf16 ? 8 : 4));
    long blocks = (total + 255) / 256;
    const long cap = (long)num_cus() * 32;
    if (blocks > cap) blocks = cap;
    hipStream_t s = (hipStream_t)stream;
    if (act_bf16) hipLaunchKernelGGL((cls_readout_fwd_kernel<bf16>), dim3((unsigned)blocks), dim3(256), 0, s, t, B, L, C, (bf16 *)out);
    else hipLaunchKernelGGL((cls_readout_fwd_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, s, t, B, L, C, (float *)out);
    return xq_check_launch(fn);
}

extern "C" int xq_cls_readout_backward(const void *g, int B, int L, int C, int act_bf16, float *gt, xq_stream_t stream) {
    const char *fn = "xq_cls_readout_backward";
    if (B < 0 || L < 1 || C < 8 || C % 8 || B > 65535) return xq_set_error(XQ_EINVAL, "%s: bad shape (C %% 8 == 0, B <= 65535)", fn);
    if (B == 0) return XQ_OK;
    if (!g || !gt) return xq_set_error(XQ_EINVAL, "%s: null pointer", fn);
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid((unsigned)B, (unsigned)((C + 63) / 64));
    if (act_bf16) hipLaunchKernelGGL((cls_readout_bwd_kernel<bf16>), grid, dim3(256), 0, s, (const bf16 *)g, L, C, gt);
    else hipLaunchKernelGGL((cls_readout_bwd_kernel<float>), grid, dim3(256), 0, s, (const float *)g, L, C, gt);
    return xq_check_launch(fn);
}

extern "C" int xq_unfold1d_circular(const void *h, int B, int L, int C, int K, int act_bf16, void *cols, xq_stream_t stream) {
    const char *fn = "xq_unfold1d_circular";
    if (int rc = fold_check(fn, B, L, C, K, act_bf16)) return rc;
    if (B == 0) return XQ_OK;
    if (!h || !cols) return xq_set_error(XQ_EINVAL, "%s: null pointer", fn);
    const long total = (long)B * L * K * (C / (act_bf16 ? 8 : 4));
    long blocks = (total + 255) / 256;
    const long cap = (long)num_cus() * 32;
    if (blocks > cap) blocks = cap;
    hipStream_t s = (hipStream_t)stream;
    if (act_bf16) hipLaunchKernelGGL((unfold1d_kernel<bf16>), dim3((unsigned)blocks), dim3(256), 0, s, (const bf16 *)h, B, L, C, K, (bf16 *)cols);
    else hipLaunchKernelGGL((unfold1d_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, s, (const float *)h, B, L, C, K, (float *)cols);
    return xq_check_launch(fn);
}

extern "C" int xq_fold1d_circular(const void *dcols, int B, int L, int C, int K, int act_bf16, void *dh, xq_stream_t stream) {
    const char *fn = "xq_fold1d_circular";
    if (int rc = fold_check(fn, B, L, C, K, act_bf16)) return rc;
    if (B == 0) return XQ_OK;
    if (!dcols || !dh) return xq_set_error(XQ_EINVAL, "%s: null pointer", fn);
    const long total = (long)B * L * (C / (act_bf16 ? 8 : 4));
    long blocks = (total + 255) / 256;
    const long cap = (long)num_cus() * 32;
    if (blocks > cap) blocks = cap;
    hipStream_t s = (hipStream_t)stream;
    if (act_bf16) hipLaunchKernelGGL((fold1d_kernel<bf16>), dim3((unsigned)blocks), dim3(256), 0, s, (const bf16 *)dcols, B, L, C, K, (bf16 *)dh);
    else hipLaunchKernelGGL((fold1d_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, s, (const float *)dcols, B, L, C, K, (float *)dh);
    return xq_check_launch(fn);
}

// ---- spectral normalisation of the head convolutions (discriminator_dino.py:121-124 -> torch.nn.utils.spectral_norm, one power
//      iteration per training forward).  The library formulation costs ~14 launches forward (two gemv, two F.normalize = norm + clamp
//      + div each, two clones, a third gemv, a dot, W / sigma) and ~10 backward per convolution — 45 convolution calls per train step.
//      Here: gemv -> vec_normalize -> gemv -> vec_normalize (its norm IS sigma: u^T W v = |W v| when u = W v / |W v|) -> W / sigma, and
//      a two-launch backward (a dot, then sn_weight_grad).  All fp32. ------------------------------------------------------------------
__global__ __launch_bounds__(256) void vec_normalize_kernel(const float *__restrict__ x, int n, float eps, float *__restrict__ out,
                                                            float *__restrict__ norm_out) {
    __shared__ float red[4];
    float s = 0.0f;
    for (int i = threadIdx.x; i < n; i += 256) s = __builtin_fmaf(x[i], x[i], s);
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float nrm = __builtin_sqrtf((red[0] + red[1]) + (red[2] + red[3]));
    const float d = nrm > eps ? nrm : eps;                 // F.normalize: x / max(|x|, eps)
    for (int i = threadIdx.x; i < n; i += 256) out[i] = x[i] / d;
    if (norm_out && threadIdx.x == 0) norm_out[0] = nrm;
}

// g_W[i][j] = g[i][j] / sigma - (dot / sigma^2) * u[i] * v[j]   (d/dW of W / sigma with sigma = u^T W v, u and v constants; dot = <g, W>)
__global__ __launch_bounds__(256) void sn_weight_grad_kernel(const float *__restrict__ g, const float *__restrict__ u, const float *__restrict__ v,
                                                             const float *__restrict__ sigma, const float *__restrict__ dot, long rows, long cols,
                                                             float *__restrict__ out) {
    const float inv = 1.0f / sigma[0];
    const float coef = dot[0] * inv * inv;
    const long n = rows * cols;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long r = i / cols, c = i - r * cols;
        out[i] = g[i] * inv - coef * u[r] * v[c];
    }
}

// ---- round 5: the same, BATCHED over H same-shaped weights.  The five DinoDisc heads hold the same three convolutions each
//      (discriminator_dino.py:209-216), so a discriminator forward normalises 3 x 5 weights of 3 shapes: per-weight that was 5 x 7 launches
//      forward and 5 x 4 backward per shape (45 convolution calls = ~430 launches of 4 - 9 us per train step, 2.7 ms).  Here one launch chain
//      per SHAPE: gemv^T over all heads -> [normalise v in the prologue] gemv over all heads -> [normalise u, sigma in the prologue] W / sigma
//      written straight in the layout (and the bf16 copy) the convolution's GEMM reads.  W: fp32 [H][R][Cin * taps] — a Conv1d weight
//      (out, in, taps) flattened as torch's spectral_norm does (dim 0 kept); out: [H][R][taps][Cin] (the unfolded convolution's
//      reduction order; taps = 1: unchanged).  Every dot is a fixed-order reduction: deterministic.
//   sn_b_gemvt : t[h][c] = sum_r W[h][r][c] u[h][r]                          grid (ceil(C / 64), H)
//   sn_b_gemvn : v = t / max(|t|, eps) (each block for itself; block 0 of a head publishes it); s[h][r] = sum_c W[h][r][c] v[c]   grid (R, H)
//   sn_b_scale : sigma = |s|, u = s / max(|s|, eps) (block 0 of a head publishes both); out = W / sigma in the GEMM layout  grid (blocks, H)
__global__ __launch_bounds__(256) void sn_b_gemvt_kernel(const float *__restrict__ W, const float *__restrict__ u, int R, int C, float *__restrict__ t) {
    // 64 columns per block; wave g walks rows g, g + 4, ... (one 256-byte row segment per load), the four partial sums meet in LDS in a fixed order
    const int h = blockIdx.y, col = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
    __shared__ float part[4][64];
    const float *Wh = W + (long)h * R * C, *uh = u + (long)h * R;
    float acc = 0.0f;
    if (col < C)
        for (int r = rg; r < R; r += 4) acc = __builtin_fmaf(Wh[(long)r * C + col], uh[r], acc);
    part[rg][threadIdx.x & 63] = acc;
    __syncthreads();
    if (threadIdx.x < 64 && col < C) t[(long)h * C + col] = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
}

// block-wide sum of squares of x[0..n) in a fixed order (thread-strided chains, wave sums, four partials)
__device__ __forceinline__ float block_sumsq_256(const float *__restrict__ x, int n, float *red /* [4] */) {
    float s = 0.0f;
    for (int i = threadIdx.x; i < n; i += 256) s = __builtin_fmaf(x[i], x[i], s);
    s = wave_sum(s);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void sn_b_gemvn_kernel(const float *__restrict__ W, const float *__restrict__ t, int R, int C, float eps,
                                                         float *__restrict__ v_buf, float *__restrict__ v_out, float *__restrict__ s) {
    const int h = blockIdx.y, r = blockIdx.x;
    __shared__ float red[4];
    const float *th = t + (long)h * C;
    const float nrm = __builtin_sqrtf(block_sumsq_256(th, C, red));
    const float d = nrm > eps ? nrm : eps;                 // F.normalize: x / max(|x|, eps)
    const float *Wr = W + ((long)h * R + r) * C;
    float acc = 0.0f;
    for (int c = threadIdx.x; c < C; c += 256) {
        const float vc = th[c] / d;
        acc = __builtin_fmaf(Wr[c], vc, acc);
        if (r == 0) { v_buf[(long)h * C + c] = vc; v_out[(long)h * C + c] = vc; }
    }
    acc = wave_sum(acc);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) s[(long)h * R + r] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void sn_b_scale_kernel(const float *__restrict__ W, const float *__restrict__ s, int R, int Cin, int taps, float eps,
                                                         float *__restrict__ u_buf, float *__restrict__ u_out, float *__restrict__ sigma,
                                                         float *__restrict__ out32, __hip_bfloat16 *__restrict__ out16) {
    const int h = blockIdx.y;
    __shared__ float red[4];
    const float *sh = s + (long)h * R;
    const float nrm = __builtin_sqrtf(block_sumsq_256(sh, R, red));
    const float d = nrm > eps ? nrm : eps;
    // u^T W v = |W v|^2 / max(|W v|, eps) = |W v| unless W v underflows eps (as the per-weight path: the norm IS sigma)
    if (blockIdx.x == 0) {
        for (int i = threadIdx.x; i < R; i += 256) { const float ui = sh[i] / d; u_buf[(long)h * R + i] = ui; u_out[(long)h * R + i] = ui; }
        if (threadIdx.x == 0) sigma[h] = nrm;
    }
    const int C = Cin * taps;
    const long n = (long)R * C, base = (long)h * n;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        // i indexes the OUTPUT [r][tap][cin]; the source is [r][cin][tap]
        const long r = i / C;
        const int rem = (int)(i - r * C), tap = rem / Cin, cin = rem - tap * Cin;
        const float val = W[base + r * C + (long)cin * taps + tap] / nrm;
        out32[base + i] = val;
        if (out16) out16[base + i] = __float2bfloat16(val);
    }
}

// backward of out = W / sigma (u, v constant): g_W = g / sigma - (<g, W> / sigma^2) u v^T.  g arrives in the OUTPUT layout [r][tap][cin], g_W leaves
// in the weight's layout [r][cin][tap].  sn_b_dot: per-block partial sums of g . W (matching elements) -> partials[h][nblk];
// sn_b_grad: every block folds its head's partials in the same order, then writes its share of g_W.
__global__ __launch_bounds__(256) void sn_b_dot_kernel(const float *__restrict__ g, const float *__restrict__ W, int R, int Cin, int taps,
                                                       float *__restrict__ partials) {
    const int h = blockIdx.y, C = Cin * taps;
    const long n = (long)R * C, base = (long)h * n;
    float acc = 0.0f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long r = i / C;
        const int rem = (int)(i - r * C), tap = rem / Cin, cin = rem - tap * Cin;
        acc = __builtin_fmaf(g[base + i], W[base + r * C + (long)cin * taps + tap], acc);
    }
    __shared__ float red[4];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partials[(long)h * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void sn_b_grad_kernel(const float *__restrict__ g, const float *__restrict__ u, const float *__restrict__ v,
                                                        const float *__restrict__ sigma, const float *__restrict__ partials, int nparts, int R, int Cin,
                                                        int taps, float *__restrict__ gW) {
    const int h = blockIdx.y, C = Cin * taps;
    __shared__ float dot_s;
    if (threadIdx.x == 0) {
        float dsum = 0.0f;
        for (int i = 0; i < nparts; ++i) dsum += partials[(long)h * nparts + i];
        dot_s = dsum;
    }
    __syncthreads();
    const float inv = 1.0f / sigma[h];
    const float coef = dot_s * inv * inv;
    const long n = (long)R * C, base = (long)h * n;
    const float *uh = u + (long)h * R, *vh = v + (long)h * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        // i indexes g_W [r][cin][tap]; the matching gradient element sits at [r][tap][cin]
        const long r = i / C;
        const int rem = (int)(i - r * C), cin = rem / taps, tap = rem - cin * taps;
        gW[base + i] = g[base + r * C + (long)tap * Cin + cin] * inv - coef * uh[r] * vh[rem];
    }
}

static int sn_b_blocks(long n) {
    long b = (n + 255) / 256;
    const long cap = (long)num_cus() * 2;
    if (b > cap) b = cap;
    return (int)(b < 1 ? 1 : b);
}

extern "C" int xq_sn_batched_workspace_floats(int H, int R, int Cin, int taps) {
    return H * (Cin * taps + R) + H * sn_b_blocks((long)R * Cin * taps);     // t, s, dot partials
}

extern "C" int xq_sn_batched_forward(const float *W, int H, int R, int Cin, int taps, float eps, float *u_buf, float *v_buf, float *u_out, float *v_out,
                                     float *sigma, float *workspace, float *out32, void *out16, xq_stream_t stream) {
    const char *fn = "xq_sn_batched_forward";
    if (H <= 0) return XQ_OK;
    if (R < 1 || Cin < 1 || taps < 1) return xq_set_error(XQ_EINVAL, "%s: bad shape", fn);
    if (!W || !u_buf || !v_buf || !u_out || !v_out || !sigma || !workspace || !out32) return xq_set_error(XQ_EINVAL, "%s: null pointer", fn);
    hipStream_t s = (hipStream_t)stream;
    const int C = Cin * taps;
    float *t = workspace, *sv = workspace + (long)H * C;
    hipLaunchKernelGGL(sn_b_gemvt_kernel, dim3((C + 63) / 64, H), dim3(256), 0, s, W, u_buf, R, C, t);
    hipLaunchKernelGGL(sn_b_gemvn_kernel, dim3(R, H), dim3(256), 0, s, W, t, R, C, eps, v_buf, v_out, sv);
    hipLaunchKernelGGL(sn_b_scale_kernel, dim3(sn_b_blocks((long)R * C), H), dim3(256), 0, s, W, sv, R, Cin, taps, eps, u_buf, u_out, sigma, out32,
                       (__hip_bfloat16 *)out16);
    return xq_check_launch(fn);
}

extern "C" int xq_sn_batched_backward(const float *g, const float *W, const float *u, const float *v, const float *sigma, int H, int R, int Cin, int taps,
                                      float *workspace, float *gW, xq_stream_t stream) {
    const char *fn = "xq_sn_batched_backward";
    if (H <= 0) return XQ_OK;
    if (R < 1 || Cin < 1 || taps < 1) return xq_set_error(XQ_EINVAL, "%s: bad shape", fn);
    if (!g || !W || !u || !v || !sigma || !workspace || !gW) return xq_set_error(XQ_EINVAL, "%s: null pointer", fn);
    hipStream_t s = (hipStream_t)stream;
    const int C = Cin * taps, nb = sn_b_blocks((long)R * C);
    float *partials = workspace + (long)H * (C + R);
    hipLaunchKernelGGL(sn_b_dot_kernel, dim3(nb, H), dim3(256), 0, s, g, W, R, Cin, taps, partials);
    hipLaunchKernelGGL(sn_b_grad_kernel, dim3(nb, H), dim3(256), 0, s, g, u, v, sigma, partials, nb, R, Cin, taps, gW);
    return xq_check_launch(fn);
}

extern "C" int xq_vec_normalize(const float *x, int n, float eps, float *out, float *norm_out, xq_stream_t stream) {
    if (n <= 0) return XQ_OK;
    if (!x || !out) return xq_set_error(XQ_EINVAL, "%s: null pointer", "xq_vec_normalize");
    hipLaunchKernelGGL(vec_normalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, x, n, eps, out, norm_out);
    return xq_check_launch("xq_vec_normalize");
}

extern "C" int xq_sn_weight_grad(const float *g, const float *u, const float *v, const float *sigma, const float *dot, int64_t rows, int64_t cols,
                                 float *out, xq_stream_t stream) {
    if (rows <= 0 || cols <= 0) return XQ_OK;
    if (!g || !u || !v || !sigma || !dot || !out) return xq_set_error(XQ_EINVAL, "%s: null pointer", "xq_sn_weight_grad");
    long blocks = (rows * cols + 255) / 256;
    const long cap = (long)num_cus() * 8;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(sn_weight_grad_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, g, u, v, sigma, dot, (long)rows, (long)cols, out);
    return xq_check_launch("xq_sn_weight_grad");
}
