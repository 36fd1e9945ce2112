// xq_gn.hip — GroupNorm (+ SiLU) on NHWC bf16 activations, forward and backward (gfx950).
//
// Replaces the `Normalize` = GroupNorm(32 groups, eps 1e-6, affine) + `nonlinearity` = x * sigmoid(x) pairs of the reference's
// CNN encoder/decoder (tokenizer/tokenizer_image/xqgan_model.py:625-640 ResnetBlock norm1/norm2, :643-672 AttnBlock.norm,
// :520-521 / :581-582 norm_out), which under bf16 autocast run as an fp32 group_norm (the input is cast up) followed by
// fp32 sigmoid and mul kernels on NCHW / strided tensors.  Here: x [B][HW][C] bf16 (channels-last, the layout of the conv
// kernels), statistics from ONE pass over x (round 5: per-channel sums of x - K_c and (x - K_c)^2, K_c = the sample's first pixel, folded to mean / rstd in fp64;
// XQ_GN_TWO_PASS=1: mean, then the centred sum of squares, rounds 2-4), y = silu((x - mean) * rstd * w + b) -> bf16.
//
//   gn_reduce_kernel<MODE> : per (sample, pixel slab): per-channel partial sums over the slab's pixels, folded to the
//                            32 groups -> partial[b][slab][G][2]  (MODE 0: sum x | 1: sum (x - mean)^2 |
//                            2: backward sums  s1 = sum d_pre * w, s2 = sum d_pre * w * xhat  + per-channel dw/db partials)
//   gn_finalize_kernel     : fixed-order sum over the slabs -> mean / rstd / (s1, s2) per (sample, group)
//   gn_apply_fwd/bwd       : element-wise passes
// A thread owns one 16-byte chunk (8 channels) of the channel axis and strides over pixels, so every wave instruction
// reads whole 128/256-byte pixel rows.
#include "xq_common.hpp"
#include "xq_internal.hpp"
#include "../../include/xq_ops.h"

#include "xq_vec.hpp"
#include <cstdlib>

using namespace xq;

static constexpr int GN_MAX_SLABS = 64;

__device__ __forceinline__ void gn_load8(const bf16 *p, float (&v)[8]) { load_vec<bf16, 8>(p, v); }

__device__ __forceinline__ float gn_silu_grad(float pre) {
    const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-pre));
    return sg * (1.0f + pre * (1.0f - sg));
}

// MODE 0: a = sum x                      (b unused)
// MODE 1: a = sum (x - mean)^2
// MODE 2: a = sum d_pre (per channel), b = sum d_pre * xhat (per channel); group partials s1 = sum_c w_c a_c, s2 = sum_c w_c b_c
// MODE 3 (round 5, the forward's statistics in ONE pass over x): a = sum (x - K_c), b = sum (x - K_c)^2 per channel, shifted by the channel's
//         value at the sample's first pixel (K_c = x[b][0][c], exact in fp32) so that neither sum carries the mean's magnitude; per-channel
//         slab sums only (part_c) — gn_finalize_onepass_kernel folds them to mean / rstd in fp64
template <int MODE>
__global__ __launch_bounds__(256) void gn_reduce_kernel(const bf16 *__restrict__ x, const bf16 *__restrict__ dy, const float *__restrict__ w,
                                                        const float *__restrict__ bias, const float *__restrict__ mean,
                                                        const float *__restrict__ rstd, int HW, int C, int G, int silu, int slab_px,
                                                        float *__restrict__ part_g, float *__restrict__ part_c) {
    __shared__ float red[2][256 * 8];      // [a | b][pixel lane * chunks + chunk][8 channels]  = per-thread partials
    __shared__ float chan[2][1024];        // per-channel slab sums (C <= 1024)
    const int cv = C / 8, PL = 256 / cv;
    const int cidx = threadIdx.x % cv, pl = threadIdx.x / cv;
    const int b = blockIdx.y, slab = blockIdx.x, nslab = gridDim.x;
    const int cg = C / G;
    const int p0 = slab * slab_px;
    int p1 = p0 + slab_px;
    if (p1 > HW) p1 = HW;
    float a[8], bq[8], mu[8], rs[8], ww[8], bb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        a[j] = 0.0f;
        bq[j] = 0.0f;
        const int c = cidx * 8 + j, g = c / cg;
        mu[j] = (MODE == 1 || MODE == 2) ? mean[b * G + g] : 0.0f;
        rs[j] = MODE == 2 ? rstd[b * G + g] : 1.0f;
        ww[j] = (MODE == 2 && w) ? w[c] : 1.0f;
        bb[j] = (MODE == 2 && bias) ? bias[c] : 0.0f;
    }
    const bf16 *xb = x + ((long)b * HW) * C + cidx * 8;
    const bf16 *db = MODE == 2 ? dy + ((long)b * HW) * C + cidx * 8 : nullptr;
    if (MODE == 3) gn_load8(xb, mu);      // the shift K_c (pixel 0 of this sample)
    for (int p = p0 + pl; p < p1; p += PL) {
        float v[8];
        gn_load8(xb + (long)p * C, v);
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] += v[j];
        } else if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = v[j] - mu[j]; a[j] = fmaf(d, d, a[j]); }
        } else if (MODE == 3) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = v[j] - mu[j]; a[j] += d; bq[j] = fmaf(d, d, bq[j]); }
        } else {
            float g8[8];
            gn_load8(db + (long)p * C, g8);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xh = (v[j] - mu[j]) * rs[j];
                const float dpre = silu ? g8[j] * gn_silu_grad(fmaf(xh, ww[j], bb[j])) : g8[j];
                a[j] += dpre;
                bq[j] = fmaf(dpre, xh, bq[j]);
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        red[0][threadIdx.x * 8 + j] = a[j];
        if (MODE >= 2) red[1][threadIdx.x * 8 + j] = bq[j];
    }
    __syncthreads();
    // per-channel sums over the pixel lanes, fixed order
    for (int c = threadIdx.x; c < C; c += 256) {
        const int ci = c >> 3, j = c & 7;
        float sa = 0.0f, sb = 0.0f;
        for (int l = 0; l < PL; ++l) {
            sa += red[0][(l * cv + ci) * 8 + j];
            if (MODE >= 2) sb += red[1][(l * cv + ci) * 8 + j];
        }
        chan[0][c] = sa;
        if (MODE >= 2) {
            chan[1][c] = sb;
            part_c[(((long)b * nslab + slab) * 2 + 0) * C + c] = sa;
            part_c[(((long)b * nslab + slab) * 2 + 1) * C + c] = sb;
        }
    }
    if (MODE == 3) return;      // (block-uniform) no group partials: the finalize kernel folds the per-channel sums
    __syncthreads();
    if (threadIdx.x < G) {
        const int g = threadIdx.x;
        float s1 = 0.0f, s2 = 0.0f;
        for (int c = g * cg; c < (g + 1) * cg; ++c) {
            const float wc = (MODE == 2 && w) ? w[c] : 1.0f;
            s1 = fmaf(wc, chan[0][c], s1);
            if (MODE == 2) s2 = fmaf(wc, chan[1][c], s2);
        }
        part_g[(((long)b * nslab + slab) * G + g) * 2 + 0] = s1;
        part_g[(((long)b * nslab + slab) * G + g) * 2 + 1] = s2;
    }
}

// what 0: mean = sum / n ; 1: rstd = 1 / sqrt(sumsq / n + eps) ; 2: s1/n, s2/n into out0 / out1
__global__ __launch_bounds__(64) void gn_finalize_kernel(const float *__restrict__ part_g, int nslab, int G, float n, float eps, int what,
                                                         float *__restrict__ out0, float *__restrict__ out1) {
    const int b = blockIdx.x, g = threadIdx.x;
    if (g >= G) return;
    float s1 = 0.0f, s2 = 0.0f;
    const float2 *pg = reinterpret_cast<const float2 *>(part_g) + (long)b * nslab * G + g;
    int s = 0;
    for (; s + 8 <= nslab; s += 8) {      // eight slabs requested before any is added, added in ascending order (same bits as one at a time; the
        float2 v[8];                      // one-at-a-time walk was a chain of 256 dependent round trips: 11 us per call)
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = pg[(long)(s + u) * G];
#pragma unroll
        for (int u = 0; u < 8; ++u) { s1 += v[u].x; s2 += v[u].y; }
    }
    for (; s < nslab; ++s) {
        const float2 v = pg[(long)s * G];
        s1 += v.x;
        s2 += v.y;
    }
    if (what == 0) out0[b * G + g] = s1 / n;
    else if (what == 1) out0[b * G + g] = 1.0f / sqrtf(s1 / n + eps);
    else { out0[b * G + g] = s1 / n; out1[b * G + g] = s2 / n; }
}

// one-pass statistics (gn_reduce_kernel<3>): S1_c = sum (x - K_c), S2_c = sum (x - K_c)^2 over the sample's HW pixels, fixed-order sums over the
// slabs; mean = sum_c (S1_c + HW K_c) / n;  sum (x - mean)^2 = sum_c [S2_c - 2 (mean - K_c) S1_c + HW (mean - K_c)^2]   (fp64: B x G threads)
// Round 6: one WAVE per (sample, group) — the 64 lanes split the slabs of every channel and meet in a xor butterfly (fp64, fixed order), instead of
// one THREAD per group walking cg x nslab x 2 strided loads one after the other (28 us per call at 256 slabs, 67 calls per step of the CNN config).
__global__ __launch_bounds__(256) void gn_finalize_onepass_kernel(const float *__restrict__ part_c, const bf16 *__restrict__ x, int nslab, int HW, int C,
                                                                  int G, float eps, float *__restrict__ mean, float *__restrict__ rstd) {
    const int b = blockIdx.x, lane = threadIdx.x & 63;
    const int g = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (g >= G) return;
    const int cg = C / G;
    const double n = (double)HW * cg;
    // lane i (< cg <= 32) keeps channel i's S1, S2, K; every lane knows the group total
    double tot = 0.0, my_a = 0.0, my_q = 0.0, my_k = 0.0;
    for (int i = 0; i < cg; ++i) {
        const int c = g * cg + i;
        double a = 0.0, q = 0.0;
        for (int s = lane; s < nslab; s += 64) {
            a += (double)part_c[(((long)b * nslab + s) * 2 + 0) * C + c];
            q += (double)part_c[(((long)b * nslab + s) * 2 + 1) * C + c];
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); q += __shfl_xor(q, o); }
        const double kc = (double)__bfloat162float(x[((long)b * HW) * C + c]);
        tot += a + (double)HW * kc;
        if (lane == i) { my_a = a; my_q = q; my_k = kc; }
    }
    const double mu = tot / n;
    const double d = mu - my_k;
    double m2 = lane < cg ? my_q - 2.0 * d * my_a + (double)HW * d * d : 0.0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m2 += __shfl_xor(m2, o);
    if (m2 < 0.0) m2 = 0.0;
    if (lane == 0) {
        mean[b * G + g] = (float)mu;
        rstd[b * G + g] = (float)(1.0 / sqrt(m2 / n + (double)eps));
    }
}

// element-wise passes: block (slab, sample); a thread keeps its 8 channels for the whole slab (as in gn_reduce_kernel), so the
// per-channel constants live in registers and the loop body is load -> 8 x (fma, [silu]) -> store — no index arithmetic per element
// (the first version derived (sample, group, channel) from a flat index with 64-bit divisions and re-read mean / rstd / w / b per
// element: 0.9 TB/s at 128 channels x 256^2, profiles/r02_cnn_b32_kernel_stats_v1.txt)
__device__ __forceinline__ float gn_sigmoid(float pre) { return __builtin_amdgcn_rcpf(1.0f + __expf(-pre)); }

__global__ __launch_bounds__(256) void gn_apply_fwd_kernel(const bf16 *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                                                           const float *__restrict__ mean, const float *__restrict__ rstd, int HW, int C,
                                                           int G, int silu, int slab_px, bf16 *__restrict__ y) {
    const int cv = C / 8, PL = 256 / cv;
    const int cidx = threadIdx.x % cv, pl = threadIdx.x / cv;
    const int b = blockIdx.y, cg = C / G;
    const int p0 = blockIdx.x * slab_px;
    int p1 = p0 + slab_px;
    if (p1 > HW) p1 = HW;
    float mu[8], rs[8], ww[8], bb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = cidx * 8 + j, g = c / cg;
        mu[j] = mean[b * G + g];
        rs[j] = rstd[b * G + g];
        ww[j] = w ? w[c] : 1.0f;
        bb[j] = bias ? bias[c] : 0.0f;
    }
    const long base = ((long)b * HW) * C + cidx * 8;
    for (int p = p0 + pl; p < p1; p += PL) {
        float v[8], o[8];
        gn_load8(x + base + (long)p * C, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float pre = fmaf((v[j] - mu[j]) * rs[j], ww[j], bb[j]);
            o[j] = silu ? pre * gn_sigmoid(pre) : pre;
        }
        store_vec<bf16, 8>(y + base + (long)p * C, o);
    }
}

__global__ __launch_bounds__(256) void gn_apply_bwd_kernel(const bf16 *__restrict__ x, const bf16 *__restrict__ dy, const float *__restrict__ w,
                                                           const float *__restrict__ bias, const float *__restrict__ mean,
                                                           const float *__restrict__ rstd, const float *__restrict__ m1,
                                                           const float *__restrict__ m2, int HW, int C, int G, int silu, int slab_px,
                                                           bf16 *__restrict__ dx) {
    const int cv = C / 8, PL = 256 / cv;
    const int cidx = threadIdx.x % cv, pl = threadIdx.x / cv;
    const int b = blockIdx.y, cg = C / G;
    const int p0 = blockIdx.x * slab_px;
    int p1 = p0 + slab_px;
    if (p1 > HW) p1 = HW;
    float mu[8], rs[8], ww[8], bb[8], a1[8], a2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = cidx * 8 + j, g = c / cg;
        mu[j] = mean[b * G + g];
        rs[j] = rstd[b * G + g];
        ww[j] = w ? w[c] : 1.0f;
        bb[j] = bias ? bias[c] : 0.0f;
        a1[j] = m1[b * G + g];
        a2[j] = m2[b * G + g];
    }
    const long base = ((long)b * HW) * C + cidx * 8;
    for (int p = p0 + pl; p < p1; p += PL) {
        float v[8], g8[8], o[8];
        gn_load8(x + base + (long)p * C, v);
        gn_load8(dy + base + (long)p * C, g8);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float xh = (v[j] - mu[j]) * rs[j];
            const float dpre = silu ? g8[j] * gn_silu_grad(fmaf(xh, ww[j], bb[j])) : g8[j];
            o[j] = rs[j] * (dpre * ww[j] - a1[j] - xh * a2[j]);
        }
        store_vec<bf16, 8>(dx + base + (long)p * C, o);
    }
}

static int gn_check(const char *fn, int B, int HW, int C, int G) {
    if (B < 0 || HW < 1 || G < 1 || G > 64 || C % G != 0 || C % 8 != 0 || C > 1024 || 256 % (C / 8) != 0 || (C / G) % 4 != 0)
        return xq_set_error(XQ_EINVAL, "%s: unsupported geometry (C=%ld, G=%ld): needs C in {64,128,256,512,1024}-like (256 %% (C/8) == 0), C/G %% 4 == 0, G <= 64",
                            fn, (long)C, (long)G);
    return XQ_OK;
}

static int gn_slabs(int HW, int C, int *slab_px) {
    const int PL = 256 / (C / 8);
    int ns = (HW + PL * 16 - 1) / (PL * 16);   // >= 16 pixels per thread
    if (ns > GN_MAX_SLABS) ns = GN_MAX_SLABS;
    if (ns < 1) ns = 1;
    *slab_px = (HW + ns - 1) / ns;
    return (HW + *slab_px - 1) / *slab_px;
}

// slabs of the element-wise passes: ~8 pixels per thread (enough independent 16-byte loads in flight per wave, >= 4 blocks per CU
// from 32 x 32 maps on)
static int gn_apply_slabs(int HW, int C, int *slab_px) {
    const int PL = 256 / (C / 8);
    *slab_px = PL * 8;
    return (HW + *slab_px - 1) / *slab_px;
}

extern "C" size_t xq_groupnorm_workspace_floats(int B, int HW, int C, int G) {
    int spx;
    const int ns = gn_slabs(HW, C, &spx);
    return (size_t)B * ns * G * 2 + (size_t)B * ns * 2 * C + 2 * (size_t)B * G;
}

extern "C" int xq_groupnorm_silu_forward(const void *x, const float *w, const float *bias, int B, int HW, int C, int G, float eps, int silu,
                                         void *y, float *mean, float *rstd, float *workspace, xq_stream_t stream) {
    const char *fn = "xq_groupnorm_silu_forward";
    if (int rc = gn_check(fn, B, HW, C, G)) return rc;
    if (B == 0) return XQ_OK;
    if (!x || !y || !mean || !rstd || !workspace) return xq_set_error(XQ_EINVAL, "%s: null pointer", fn);
    hipStream_t s = (hipStream_t)stream;
    int spx;
    const int ns = gn_slabs(HW, C, &spx);
    const float n = (float)HW * (float)(C / G);
    const bf16 *xp = (const bf16 *)x;
    const int pslot = xq::prof_begin(XQ_PROF_GROUPNORM, (double)B * HW * C * 8.0, s);
    static const bool two_pass = getenv("XQ_GN_TWO_PASS") != nullptr;      // rounds 2-4: mean, then the centred sum of squares (two reads of x)
    if (!two_pass && C / G <= 32) {
        float *part_c = workspace + (size_t)B * ns * G * 2;
        hipLaunchKernelGGL((gn_reduce_kernel<3>), dim3(ns, B), dim3(256), 0, s, xp, (const bf16 *)nullptr, w, bias, mean, rstd, HW, C, G, silu, spx,
                           workspace, part_c);
        hipLaunchKernelGGL(gn_finalize_onepass_kernel, dim3(B, (G + 3) / 4), dim3(256), 0, s, (const float *)part_c, xp, ns, HW, C, G, eps, mean, rstd);
    } else {
        hipLaunchKernelGGL((gn_reduce_kernel<0>), dim3(ns, B), dim3(256), 0, s, xp, (const bf16 *)nullptr, w, bias, mean, rstd, HW, C, G, silu, spx,
                           workspace, (float *)nullptr);
        hipLaunchKernelGGL(gn_finalize_kernel, dim3(B), dim3(64), 0, s, workspace, ns, G, n, eps, 0, mean, (float *)nullptr);
        hipLaunchKernelGGL((gn_reduce_kernel<1>), dim3(ns, B), dim3(256), 0, s, xp, (const bf16 *)nullptr, w, bias, mean, rstd, HW, C, G, silu, spx,
                           workspace, (float *)nullptr);
        hipLaunchKernelGGL(gn_finalize_kernel, dim3(B), dim3(64), 0, s, workspace, ns, G, n, eps, 1, rstd, (float *)nullptr);
    }
    int apx;
    const int ans = gn_apply_slabs(HW, C, &apx);
    hipLaunchKernelGGL(gn_apply_fwd_kernel, dim3(ans, B), dim3(256), 0, s, xp, w, bias, mean, rstd, HW, C, G, silu, apx, (bf16 *)y);
    xq::prof_end(pslot, s);
    return xq_check_launch(fn);
}

// g_wb_partials: fp32 [B * nslab][2][C] (row 0: d bias partials = sum d_pre, row 1: d weight partials = sum d_pre * xhat);
// the caller sums them over the first axis; *n_partial_rows receives B * nslab.
extern "C" int xq_groupnorm_silu_backward(const void *x, const void *dy, const float *w, const float *bias, const float *mean,
                                          const float *rstd, int B, int HW, int C, int G, int silu, void *dx, float *g_wb_partials,
                                          int *n_partial_rows, float *workspace, xq_stream_t stream) {
    const char *fn = "xq_groupnorm_silu_backward";
    if (int rc = gn_check(fn, B, HW, C, G)) return rc;
    if (n_partial_rows) *n_partial_rows = 0;
    if (B == 0) return XQ_OK;
    if (!x || !dy || !mean || !rstd || !dx || !g_wb_partials || !workspace) return xq_set_error(XQ_EINVAL, "%s: null pointer", fn);
    hipStream_t s = (hipStream_t)stream;
    int spx;
    const int ns = gn_slabs(HW, C, &spx);
    if (n_partial_rows) *n_partial_rows = B * ns;
    const float n = (float)HW * (float)(C / G);
    float *part_g = workspace;                               // [B][ns][G][2]
    float *m1 = workspace + (size_t)B * ns * G * 2 + (size_t)B * ns * 2 * C;   // after the (unused here) per-channel area
    float *m2 = m1 + (size_t)B * G;
    const int pslot = xq::prof_begin(XQ_PROF_GROUPNORM, (double)B * HW * C * 10.0, s);
    hipLaunchKernelGGL((gn_reduce_kernel<2>), dim3(ns, B), dim3(256), 0, s, (const bf16 *)x, (const bf16 *)dy, w, bias, mean, rstd, HW, C, G, silu,
                       spx, part_g, g_wb_partials);
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(B), dim3(64), 0, s, part_g, ns, G, n, 0.0f, 2, m1, m2);
    int apx;
    const int ans = gn_apply_slabs(HW, C, &apx);
    hipLaunchKernelGGL(gn_apply_bwd_kernel, dim3(ans, B), dim3(256), 0, s, (const bf16 *)x, (const bf16 *)dy, w, bias, mean, rstd, m1, m2, HW, C, G,
                       silu, apx, (bf16 *)dx);
    xq::prof_end(pslot, s);
    return xq_check_launch(fn);
}
