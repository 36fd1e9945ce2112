// xq_aug.hip — DiffAug (translation, colour, cut-out) on a planar fp32 image batch in two launches per direction (gfx950).
//
// Replaces the ~15 elementwise / gather / reduction launches per call of the reference's differentiable augmentation
// (tokenizer/tokenizer_image/diffaug.py:64-118; called three times per train step on (B, 3, 256, 256) images: vq_loss.py:169,209-210)
// and their ~20 autograd launches in the generator pass.  Per sample b, with the seven uniform draws r0..r6 of the call
// (rand01[k][b], drawn by torch exactly as upstream, so the random stream is unchanged):
//   translation : x0[c][i][j] = x[c][i + th][j + tw] inside the image, 0 outside;  th = floor(r0 (2 dh + 1)) - dh,  tw likewise (r1)
//   brightness  : x1 = x0 + (r2 - 0.5)
//   saturation  : x2 = (x1 - m1) (2 r3) + m1,      m1 = mean over the 3 channels of the pixel
//   contrast    : x3 = (x2 - m2) (r4 + 0.5) + m2,  m2 = mean over (c, i, j) of x2  = mean(x0) + (r2 - 0.5)  (saturation keeps pixel means)
//   cut-out     : x4 = x3 * [ (i, j) outside the ch x cw box centred at (floor(r5 (H + 1 - ch % 2)), floor(r6 (W + 1 - cw % 2))) ]
// Pass 1 reduces the per-sample sum (of x over the translated window forward, of g * mask backward) into per-block partials
// (summed in a fixed order by pass 2: deterministic); pass 2 is one thread per pixel, three planes, coalesced rows.
// The backward is the exact transpose: g3 = g mask; g2 = con g3 + (1 - con) sum(g3) / (3 H W); g1 = sat g2 + (1 - sat) mean_c(g2);
// g_x[c][i'][j'] = g1[c][i' - th][j' - tw] inside the image.
#include "xq_common.hpp"
#include "xq_internal.hpp"
#include "../../include/xq_ops.h"

#include <hip/hip_bf16.h>

using namespace xq;

namespace {

constexpr int AUG_SLABS = 64;     // partial sums per sample

struct AugArgs {
    const float *rand01;   // [7][B]
    int B, H, W, dh, dw, ch, cw;
    int trans, color, cut;
};

struct AugSample {
    int th, tw, oh, ow;
    float br, sat, con;
};

__device__ __forceinline__ AugSample aug_sample(const AugArgs &a, int b) {
    AugSample s;
    const float *r = a.rand01;
    s.th = a.trans ? (int)floorf(r[0 * a.B + b] * (float)(2 * a.dh + 1)) - a.dh : 0;
    s.tw = a.trans ? (int)floorf(r[1 * a.B + b] * (float)(2 * a.dw + 1)) - a.dw : 0;
    s.br = r[2 * a.B + b] - 0.5f;
    s.sat = r[3 * a.B + b] * 2.0f;
    s.con = r[4 * a.B + b] + 0.5f;
    s.oh = (int)floorf(r[5 * a.B + b] * (float)(a.H + (1 - a.ch % 2)));
    s.ow = (int)floorf(r[6 * a.B + b] * (float)(a.W + (1 - a.cw % 2)));
    return s;
}

__device__ __forceinline__ float aug_mask(const AugArgs &a, const AugSample &s, int i, int j) {
    if (!a.cut) return 1.0f;
    const int sh = s.oh - a.ch / 2, sw = s.ow - a.cw / 2;
    const bool in = i >= sh && i <= sh + a.ch - 1 && j >= sw && j <= sw + a.cw - 1;
    return in ? 0.0f : 1.0f;
}

__device__ __forceinline__ float block_sum_256(float v, float *red) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// forward pass 1: partial[b][slab] = sum over the slab's DESTINATION pixels of x0 (3 channels); backward: of g * mask
template <bool BWD>
__global__ __launch_bounds__(256) void aug_sum_kernel(const float *__restrict__ x, AugArgs a, float *__restrict__ partial) {
    __shared__ float red[4];
    const int b = blockIdx.y, slab = blockIdx.x;
    const AugSample s = aug_sample(a, b);
    const long HW = (long)a.H * a.W;
    const long per = (HW + AUG_SLABS - 1) / AUG_SLABS;
    const long p0 = slab * per, p1 = p0 + per < HW ? p0 + per : HW;
    const float *xb = x + (long)b * 3 * HW;
    float acc = 0.0f;
    for (long p = p0 + threadIdx.x; p < p1; p += 256) {
        const int i = (int)(p / a.W), j = (int)(p - (long)i * a.W);
        if (BWD) {
            const float m = aug_mask(a, s, i, j);
            acc += m * ((xb[p] + xb[HW + p]) + xb[2 * HW + p]);
        } else {
            const int si = i + s.th, sj = j + s.tw;
            if (si >= 0 && si < a.H && sj >= 0 && sj < a.W) {
                const long q = (long)si * a.W + sj;
                acc += (xb[q] + xb[HW + q]) + xb[2 * HW + q];
            }
        }
    }
    const float t = block_sum_256(acc, red);
    if (threadIdx.x == 0) partial[b * AUG_SLABS + slab] = t;
}

__device__ __forceinline__ float aug_total(const float *partial, int b) {
    float t = 0.0f;
    for (int k = 0; k < AUG_SLABS; ++k) t += partial[b * AUG_SLABS + k];
    return t;
}

__global__ __launch_bounds__(256) void aug_apply_fwd_kernel(const float *__restrict__ x, AugArgs a, const float *__restrict__ partial,
                                                            float *__restrict__ y) {
    const int b = blockIdx.y;
    const AugSample s = aug_sample(a, b);
    const long HW = (long)a.H * a.W;
    const float m2 = aug_total(partial, b) / (float)(3 * HW) + s.br;
    const float *xb = x + (long)b * 3 * HW;
    float *yb = y + (long)b * 3 * HW;
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < HW; p += (long)gridDim.x * 256) {
        const int i = (int)(p / a.W), j = (int)(p - (long)i * a.W);
        const int si = i + s.th, sj = j + s.tw;
        float v0 = 0.0f, v1 = 0.0f, v2 = 0.0f;
        if (si >= 0 && si < a.H && sj >= 0 && sj < a.W) {
            const long q = (long)si * a.W + sj;
            v0 = xb[q]; v1 = xb[HW + q]; v2 = xb[2 * HW + q];
        }
        if (a.color) {
            v0 += s.br; v1 += s.br; v2 += s.br;
            const float m1 = ((v0 + v1) + v2) / 3.0f;
            v0 = (v0 - m1) * s.sat + m1; v1 = (v1 - m1) * s.sat + m1; v2 = (v2 - m1) * s.sat + m1;
            v0 = (v0 - m2) * s.con + m2; v1 = (v1 - m2) * s.con + m2; v2 = (v2 - m2) * s.con + m2;
        }
        const float m = aug_mask(a, s, i, j);
        yb[p] = v0 * m; yb[HW + p] = v1 * m; yb[2 * HW + p] = v2 * m;
    }
}

__global__ __launch_bounds__(256) void aug_apply_bwd_kernel(const float *__restrict__ g, AugArgs a, const float *__restrict__ partial,
                                                            float *__restrict__ gx) {
    const int b = blockIdx.y;
    const AugSample s = aug_sample(a, b);
    const long HW = (long)a.H * a.W;
    const float gmean = aug_total(partial, b) / (float)(3 * HW);       // mean over (c, i, j) of g * mask
    const float *gb = g + (long)b * 3 * HW;
    float *ob = gx + (long)b * 3 * HW;
    for (long p = (long)blockIdx.x * 256 + threadIdx.x; p < HW; p += (long)gridDim.x * 256) {
        const int si = (int)(p / a.W), sj = (int)(p - (long)si * a.W);     // SOURCE pixel of the forward
        const int i = si - s.th, j = sj - s.tw;                             // its destination
        float v0 = 0.0f, v1 = 0.0f, v2 = 0.0f;
        if (i >= 0 && i < a.H && j >= 0 && j < a.W) {
            const long q = (long)i * a.W + j;
            const float m = aug_mask(a, s, i, j);
            v0 = gb[q] * m; v1 = gb[HW + q] * m; v2 = gb[2 * HW + q] * m;
            if (a.color) {
                const float k = (1.0f - s.con) * gmean;
                v0 = s.con * v0 + k; v1 = s.con * v1 + k; v2 = s.con * v2 + k;
                const float mc = (1.0f - s.sat) * (((v0 + v1) + v2) / 3.0f);
                v0 = s.sat * v0 + mc; v1 = s.sat * v1 + mc; v2 = s.sat * v2 + mc;
            }
        }
        ob[p] = v0; ob[HW + p] = v1; ob[2 * HW + p] = v2;
    }
}

int aug_check(const char *fn, const void *x, const float *rand01, const void *y, const float *ws, int B, int H, int W) {
    if (B < 0 || H < 1 || W < 1) return xq_set_error(XQ_EINVAL, "%s: bad shape", fn);
    if (B > 0 && (!x || !rand01 || !y || !ws)) return xq_set_error(XQ_EINVAL, "%s: null pointer", fn);
    if (B > 65535) return xq_set_error(XQ_EINVAL, "%s: batch > 65535", fn);
    return XQ_OK;
}

}  // namespace

extern "C" size_t xq_diffaug_workspace_floats(int B) { return (size_t)(B > 0 ? B : 0) * AUG_SLABS; }

extern "C" int xq_diffaug_forward(const float *x, const float *rand01, int B, int H, int W, int dh, int dw, int ch, int cw, int trans, int color,
                                  int cut, float *y, float *workspace, xq_stream_t stream) {
    const char *fn = "xq_diffaug_forward";
    if (int rc = aug_check(fn, x, rand01, y, workspace, B, H, W)) return rc;
    if (B == 0) return XQ_OK;
    hipStream_t s = (hipStream_t)stream;
    const AugArgs a{rand01, B, H, W, dh, dw, ch, cw, trans, color, cut};
    if (color) hipLaunchKernelGGL((aug_sum_kernel<false>), dim3(AUG_SLABS, B), dim3(256), 0, s, x, a, workspace);
    long bx = ((long)H * W + 255) / 256;
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(aug_apply_fwd_kernel, dim3((unsigned)bx, B), dim3(256), 0, s, x, a, workspace, y);
    return xq_check_launch(fn);
}

extern "C" int xq_diffaug_backward(const float *g, const float *rand01, int B, int H, int W, int dh, int dw, int ch, int cw, int trans, int color,
                                   int cut, float *gx, float *workspace, xq_stream_t stream) {
    const char *fn = "xq_diffaug_backward";
    if (int rc = aug_check(fn, g, rand01, gx, workspace, B, H, W)) return rc;
    if (B == 0) return XQ_OK;
    hipStream_t s = (hipStream_t)stream;
    const AugArgs a{rand01, B, H, W, dh, dw, ch, cw, trans, color, cut};
    if (color) hipLaunchKernelGGL((aug_sum_kernel<true>), dim3(AUG_SLABS, B), dim3(256), 0, s, g, a, workspace);
    long bx = ((long)H * W + 255) / 256;
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(aug_apply_bwd_kernel, dim3((unsigned)bx, B), dim3(256), 0, s, g, a, workspace, gx);
    return xq_check_launch(fn);
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// Round 5: input side of the frozen DINO-S trunk of the discriminator (discriminator_dino.py:327-337 + its PatchEmbed :262-276) in one pass
// per direction.  Upstream: x_scale * x + x_shift (ImageNet normalisation of the [-1, 1] image), then a random 224-crop OR
// F.interpolate(size = 224, mode = 'area'), then the 16 x 16 patch convolution — here a GEMM over patchified pixels, so the chain was
// mul, add_, adaptive_avg_pool2d (its backward: an atomicAdd scatter, 0.38 ms) / a crop copy, the patchify permute-copy and a bf16 cast:
// five passes over a 100 MB image batch per call, three calls per train step.  dino_prep_fwd writes the bf16 patch matrix
// cols[(b, gy, gx)][(c, dy, dx)] directly; dino_prep_bwd turns the GEMM's data gradient back into the image gradient by GATHER (every input
// pixel sums the <= 2 x 2 output windows that cover it: deterministic).  Area windows as ATen's adaptive pooling: [floor(o H / S), ceil((o + 1) H / S)).
// ---------------------------------------------------------------------------------------------------------------------------------------
struct PrepArgs {
    int B, H, W, S, P, G;      // image B x 3 x H x W; S x S crop / resize; P x P patches, G = S / P per side
    int mode, oi, oj;          // 0: crop at (oi, oj); 1: area resize
    float scale[3], shift[3];
};

__device__ __forceinline__ void area_win(int o, int in, int out, int &lo, int &hi) {      // (o + 1) * in < 2^31: checked by the launcher
    lo = (o * in) / out;
    hi = ((o + 1) * in + out - 1) / out;
}

__global__ __launch_bounds__(256) void dino_prep_fwd_kernel(const float *__restrict__ x, PrepArgs a, __hip_bfloat16 *__restrict__ cols) {
    // one thread = 8 consecutive pixels of one row of the S x S crop / resized image; consecutive threads walk along the row, so the image
    // reads are coalesced (the 16-byte patch-matrix stores of a row land 2 per patch, 3 P P bf16 apart — they merge in L2 with the other
    // rows of the patch).  (First version: consecutive threads walked (dx chunk, dy, c) of one patch — 32-byte image reads, 132 us per call.)
    const int S8 = a.S / 8, K = 3 * a.P * a.P;
    const long total = (long)a.B * 3 * a.S * S8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int x8 = (int)(i % S8);
        long t = i / S8;
        const int Y = (int)(t % a.S); t /= a.S;
        const int c = (int)(t % 3);
        const long b = t / 3;
        const int X0 = 8 * x8;
        const float *xp = x + (b * 3 + c) * (long)a.H * a.W;
        float v[8];
        if (a.mode == 0) {
            const float *row = xp + (long)(Y + a.oi) * a.W + X0 + a.oj;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = row[e];
        } else {
            int ylo, yhi;
            area_win(Y, a.H, a.S, ylo, yhi);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                int xlo, xhi;
                area_win(X0 + e, a.W, a.S, xlo, xhi);
                float acc = 0.0f;
                for (int yy = ylo; yy < yhi; ++yy)
                    for (int xx = xlo; xx < xhi; ++xx) acc += xp[(long)yy * a.W + xx];
                v[e] = acc / (float)((yhi - ylo) * (xhi - xlo));
            }
        }
        struct alignas(16) B8 { __hip_bfloat16 h[8]; } o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o.h[e] = __float2bfloat16(__builtin_fmaf(a.scale[c], v[e], a.shift[c]));
        const int gy = Y / a.P, dy = Y - gy * a.P, gx = X0 / a.P, dx = X0 - gx * a.P;      // P % 8 == 0: the 8 pixels lie in one patch
        *reinterpret_cast<B8 *>(cols + ((b * a.G + gy) * a.G + gx) * (long)K + (c * a.P + dy) * a.P + dx) = o;
    }
}

__global__ __launch_bounds__(256) void dino_prep_bwd_kernel(const __hip_bfloat16 *__restrict__ gcols, PrepArgs a, float *__restrict__ gx) {
    const long total = (long)a.B * 3 * a.H * a.W;
    const int K = 3 * a.P * a.P;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ix = (int)(i % a.W);
        long t = i / a.W;
        const int iy = (int)(t % a.H); t /= a.H;
        const int c = (int)(t % 3);
        const long b = t / 3;
        float acc = 0.0f;
        const __hip_bfloat16 *gb = gcols + b * (long)a.G * a.G * K + (long)c * a.P * a.P;
        auto g_at = [&](int Y, int X) -> float {       // gradient of crop / resized pixel (Y, X) of plane c
            const int gy = Y / a.P, dy = Y - gy * a.P, gxx = X / a.P, dx = X - gxx * a.P;
            return __bfloat162float(gb[(long)(gy * a.G + gxx) * K + dy * a.P + dx]);
        };
        if (a.mode == 0) {
            const int Y = iy - a.oi, X = ix - a.oj;
            if (Y >= 0 && Y < a.S && X >= 0 && X < a.S) acc = g_at(Y, X);
        } else {
            // the (<= 3 candidate, <= 2 live) output rows / columns whose window holds iy / ix, with the reciprocal window lengths: the area
            // weight 1 / (wh * ww) is separable, so the two axes are resolved once each instead of per (row, column) pair
            int oyv[3], oxv[3];
            float wyv[3], wxv[3];
            const int oy0 = (iy * a.S) / a.H, ox0 = (ix * a.S) / a.W;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                int o = oy0 - 1 + k, lo, hi;
                wyv[k] = 0.0f; oyv[k] = 0;
                if (o >= 0 && o < a.S) {
                    area_win(o, a.H, a.S, lo, hi);
                    if (iy >= lo && iy < hi) { wyv[k] = 1.0f / (float)(hi - lo); oyv[k] = o; }
                }
                o = ox0 - 1 + k;
                wxv[k] = 0.0f; oxv[k] = 0;
                if (o >= 0 && o < a.S) {
                    area_win(o, a.W, a.S, lo, hi);
                    if (ix >= lo && ix < hi) { wxv[k] = 1.0f / (float)(hi - lo); oxv[k] = o; }
                }
            }
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    if (wyv[p] != 0.0f && wxv[q] != 0.0f) acc = __builtin_fmaf(wyv[p] * wxv[q], g_at(oyv[p], oxv[q]), acc);
        }
        gx[i] = acc * a.scale[c];
    }
}

static int prep_args(const char *fn, int B, int H, int W, int S, int P, int mode, int oi, int oj, const float *scale3, const float *shift3, PrepArgs *a) {
    if (B < 0 || H < 1 || W < 1 || S < 1 || P < 8 || P % 8 || S % P) return xq_set_error(XQ_EINVAL, "%s: bad geometry (patch %% 8, size %% patch)", fn);
    if (mode == 0 && (oi < 0 || oj < 0 || oi + S > H || oj + S > W)) return xq_set_error(XQ_EINVAL, "%s: crop outside the image", fn);
    if (mode == 1 && (S > H || S > W || 2L * S < H || 2L * S < W))
        return xq_set_error(XQ_EINVAL, "%s: area mode handles down-scaling by less than 2 (windows of <= 2 x 2 pixels: the backward visits 3 x 3 candidates)", fn);
    if (mode != 0 && mode != 1) return xq_set_error(XQ_EINVAL, "%s: mode 0 (crop) or 1 (area)", fn);
    if ((long)(S + 1) * (H > W ? H : W) >= 0x7fffffffL) return xq_set_error(XQ_EINVAL, "%s: image too large for 32-bit window arithmetic", fn);
    if (!scale3 || !shift3) return xq_set_error(XQ_EINVAL, "%s: null scale / shift (HOST pointers, 3 floats each)", fn);
    a->B = B; a->H = H; a->W = W; a->S = S; a->P = P; a->G = S / P; a->mode = mode; a->oi = oi; a->oj = oj;
    for (int c = 0; c < 3; ++c) { a->scale[c] = scale3[c]; a->shift[c] = shift3[c]; }
    return XQ_OK;
}

extern "C" int xq_dino_prep_patches_forward(const float *x, int B, int H, int W, int S, int P, int mode, int oi, int oj, const float *scale3_host,
                                            const float *shift3_host, void *cols_bf16, xq_stream_t stream) {
    const char *fn = "xq_dino_prep_patches_forward";
    PrepArgs a;
    if (int rc = prep_args(fn, B, H, W, S, P, mode, oi, oj, scale3_host, shift3_host, &a)) return rc;
    if (B == 0) return XQ_OK;
    if (!x || !cols_bf16) return xq_set_error(XQ_EINVAL, "%s: null pointer", fn);
    const long total = (long)B * 3 * S * (S / 8);
    long blocks = (total + 255) / 256;
    const long cap = (long)num_cus() * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(dino_prep_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, a, (__hip_bfloat16 *)cols_bf16);
    return xq_check_launch(fn);
}

extern "C" int xq_dino_prep_patches_backward(const void *gcols_bf16, int B, int H, int W, int S, int P, int mode, int oi, int oj,
                                             const float *scale3_host, const float *shift3_host, float *gx, xq_stream_t stream) {
    const char *fn = "xq_dino_prep_patches_backward";
    PrepArgs a;
    if (int rc = prep_args(fn, B, H, W, S, P, mode, oi, oj, scale3_host, shift3_host, &a)) return rc;
    if (B == 0) return XQ_OK;
    if (!gcols_bf16 || !gx) return xq_set_error(XQ_EINVAL, "%s: null pointer", fn);
    const long total = (long)B * 3 * H * W;
    long blocks = (total + 255) / 256;
    const long cap = (long)num_cus() * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(dino_prep_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const __hip_bfloat16 *)gcols_bf16, a, gx);
    return xq_check_launch(fn);
}

// out (bf16 planar) = scale_c * x + shift_c for a (B, 3, H, W) fp32 image batch: the input scaling of LPIPS ((x - shift) / scale, lpips.py:59-64)
// and the cast autocast applies in front of the first VGG convolution, in one pass (were sub, div, cast: three); backward g_x = scale_c * g.
struct alignas(8) Bf4 { __hip_bfloat16 a, b, c, d; };
__device__ __forceinline__ float4 load4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ float4 load4(const __hip_bfloat16 *p) {
    const Bf4 v = *reinterpret_cast<const Bf4 *>(p);
    return make_float4(__bfloat162float(v.a), __bfloat162float(v.b), __bfloat162float(v.c), __bfloat162float(v.d));
}
__device__ __forceinline__ void store4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
__device__ __forceinline__ void store4(__hip_bfloat16 *p, float4 v) {
    const Bf4 o = {__float2bfloat16(v.x), __float2bfloat16(v.y), __float2bfloat16(v.z), __float2bfloat16(v.w)};
    *reinterpret_cast<Bf4 *>(p) = o;
}

template <typename TI>
__global__ __launch_bounds__(256) void image_affine_fwd_kernel(const TI *__restrict__ x, long plane4, float s0, float s1, float s2, float h0,
                                                               float h1, float h2, __hip_bfloat16 *__restrict__ out, long total4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
        const int c = (int)((i / plane4) % 3);
        const float sc = c == 0 ? s0 : (c == 1 ? s1 : s2), sh = c == 0 ? h0 : (c == 1 ? h1 : h2);
        const float4 v = load4(x + i * 4);
        struct alignas(8) B4 { __hip_bfloat16 a, b, c, d; } o = {__float2bfloat16(__builtin_fmaf(sc, v.x, sh)), __float2bfloat16(__builtin_fmaf(sc, v.y, sh)),
                                                                 __float2bfloat16(__builtin_fmaf(sc, v.z, sh)), __float2bfloat16(__builtin_fmaf(sc, v.w, sh))};
        *reinterpret_cast<B4 *>(out + i * 4) = o;
    }
}

template <typename TO>
__global__ __launch_bounds__(256) void image_affine_bwd_kernel(const __hip_bfloat16 *__restrict__ g, long plane4, float s0, float s1, float s2,
                                                               TO *__restrict__ gx, long total4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
        const int c = (int)((i / plane4) % 3);
        const float sc = c == 0 ? s0 : (c == 1 ? s1 : s2);
        const float4 v = load4(g + i * 4);
        store4(gx + i * 4, make_float4(sc * v.x, sc * v.y, sc * v.z, sc * v.w));
    }
}

extern "C" int xq_image_affine_bf16_forward(const void *x, int x_is_bf16, int B, int H, int W, const float *scale3_host, const float *shift3_host, void *out_bf16,
                                            xq_stream_t stream) {
    const char *fn = "xq_image_affine_bf16_forward";
    if (B < 0 || H < 1 || W < 1 || ((long)H * W) % 4) return xq_set_error(XQ_EINVAL, "%s: bad geometry (H * W %% 4)", fn);
    if (B == 0) return XQ_OK;
    if (!x || !out_bf16 || !scale3_host || !shift3_host) return xq_set_error(XQ_EINVAL, "%s: null pointer", fn);
    const long plane4 = (long)H * W / 4, total4 = plane4 * 3 * B;
    long blocks = (total4 + 255) / 256;
    const long cap = (long)num_cus() * 16;
    if (blocks > cap) blocks = cap;
    if (x_is_bf16)
        hipLaunchKernelGGL((image_affine_fwd_kernel<__hip_bfloat16>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const __hip_bfloat16 *)x, plane4,
                           scale3_host[0], scale3_host[1], scale3_host[2], shift3_host[0], shift3_host[1], shift3_host[2], (__hip_bfloat16 *)out_bf16, total4);
    else
        hipLaunchKernelGGL((image_affine_fwd_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const float *)x, plane4,
                           scale3_host[0], scale3_host[1], scale3_host[2], shift3_host[0], shift3_host[1], shift3_host[2], (__hip_bfloat16 *)out_bf16, total4);
    return xq_check_launch(fn);
}

extern "C" int xq_image_affine_bf16_backward(const void *g_bf16, int B, int H, int W, const float *scale3_host, void *gx, int gx_is_bf16,
                                             xq_stream_t stream) {
    const char *fn = "xq_image_affine_bf16_backward";
    if (B < 0 || H < 1 || W < 1 || ((long)H * W) % 4) return xq_set_error(XQ_EINVAL, "%s: bad geometry (H * W %% 4)", fn);
    if (B == 0) return XQ_OK;
    if (!g_bf16 || !gx || !scale3_host) return xq_set_error(XQ_EINVAL, "%s: null pointer", fn);
    const long plane4 = (long)H * W / 4, total4 = plane4 * 3 * B;
    long blocks = (total4 + 255) / 256;
    const long cap = (long)num_cus() * 16;
    if (blocks > cap) blocks = cap;
    if (gx_is_bf16)
        hipLaunchKernelGGL((image_affine_bwd_kernel<__hip_bfloat16>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const __hip_bfloat16 *)g_bf16, plane4,
                           scale3_host[0], scale3_host[1], scale3_host[2], (__hip_bfloat16 *)gx, total4);
    else
        hipLaunchKernelGGL((image_affine_bwd_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const __hip_bfloat16 *)g_bf16, plane4,
                           scale3_host[0], scale3_host[1], scale3_host[2], (float *)gx, total4);
    return xq_check_launch(fn);
}
