// xq_optim.hip — fused AdamW + EMA step over flat parameter arenas (gfx950).
//
// Replaces, for the tokenizer's ~172 M trainable parameters, the per-tensor loops of the reference train step:
//   torch.optim.AdamW(...).step()          tokenizer/tokenizer_image/xqgan_train.py:344-347,459
//   update_ema(ema, model, decay=0.9999)   utils/ema.py:5-14, xqgan_train.py:461-462
//   optimizer.zero_grad()                  xqgan_train.py:447
//   the 1/world_size of DDP's gradient mean (folded in as grad_scale)
// (+ optional bf16 shadow copy of p for the GEMMs)
// One elementwise pass: reads p, g, m, v, ema (20 B/param), writes p, m, v, ema (+g = 0) (16-20 B/param):
// HBM-bound, 36-40 algorithmic bytes per parameter.  float4 accesses, grid-stride.
#include "xq_common.hpp"
#include "xq_internal.hpp"
#include "../../include/xq_ops.h"

#include <math.h>
#include <hip/hip_bf16.h>

struct AdamArgs {
    float lr, beta1, beta2, eps, weight_decay, step_size, inv_bc2_sqrt, ema_decay, grad_scale;
    int zero_grad, has_ema;
    const float *coeffs;   // device [2] = {step_size, inv_bc2_sqrt} (xq_adamw_ema_step_dev: the step count lives on the device so that
                           // the launch can sit in a hipGraph and still see an advancing step) or null
    const float *clip;     // device [2] = {total gradient norm, clip coefficient <= 1} written by xq_grad_norm_clip, or null:
                           // torch.nn.utils.clip_grad_norm_ (xqgan_train.py:456-458,471-473) without a host read — the step multiplies
                           // grad_scale by clip[1]
};

__device__ __forceinline__ void adam1(float &p, float &g, float &m, float &v, float &e, const AdamArgs &a) {
    const float gg = g * a.grad_scale;
    p = p * (1.0f - a.lr * a.weight_decay);                // decoupled weight decay (torch AdamW: param.mul_(1 - lr*wd))
    m = m + (gg - m) * (1.0f - a.beta1);                    // exp_avg.lerp_(grad, 1 - beta1)
    v = v * a.beta2 + (gg * gg) * (1.0f - a.beta2);         // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1-beta2)
    const float denom = __builtin_sqrtf(v) * a.inv_bc2_sqrt + a.eps;
    p = p - a.step_size * (m / denom);                      // param.addcdiv_(exp_avg, denom, value=-step_size)
    if (a.has_ema) e = e * a.ema_decay + p * (1.0f - a.ema_decay);  // ema.mul_(decay).add_(param, alpha=1-decay)
    if (a.zero_grad) g = 0.0f;
}

__global__ __launch_bounds__(256) void adamw_ema_kernel(float *__restrict__ p, float *__restrict__ g, float *__restrict__ m,
                                                        float *__restrict__ v, float *__restrict__ ema,
                                                        __hip_bfloat16 *__restrict__ p16, long n, AdamArgs a) {
    if (a.coeffs) { a.step_size = a.coeffs[0]; a.inv_bc2_sqrt = a.coeffs[1]; }
    if (a.clip) a.grad_scale *= a.clip[1];
    const long n4 = n >> 2;
    const long stride = (long)gridDim.x * 256;
    float4 *p4 = reinterpret_cast<float4 *>(p), *g4 = reinterpret_cast<float4 *>(g), *m4 = reinterpret_cast<float4 *>(m),
           *v4 = reinterpret_cast<float4 *>(v), *e4 = reinterpret_cast<float4 *>(ema);
    // two float4 per trip: ten 16-byte loads in flight per thread before the first is consumed (round 5: one per trip ran at 4.75 TB/s)
    for (long i0 = (long)blockIdx.x * 256 + threadIdx.x; i0 < n4; i0 += 2 * stride) {
        const long i1 = i0 + stride;
        const bool two = i1 < n4;
        const long j1 = two ? i1 : i0;
        float4 pp[2] = {p4[i0], p4[j1]}, gg[2] = {g4[i0], g4[j1]}, mm[2] = {m4[i0], m4[j1]}, vv[2] = {v4[i0], v4[j1]};
        float4 ee[2];
        ee[0] = a.has_ema ? e4[i0] : make_float4(0, 0, 0, 0);
        ee[1] = a.has_ema ? e4[j1] : make_float4(0, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u == 1 && !two) break;
            const long i = u ? i1 : i0;
            adam1(pp[u].x, gg[u].x, mm[u].x, vv[u].x, ee[u].x, a);
            adam1(pp[u].y, gg[u].y, mm[u].y, vv[u].y, ee[u].y, a);
            adam1(pp[u].z, gg[u].z, mm[u].z, vv[u].z, ee[u].z, a);
            adam1(pp[u].w, gg[u].w, mm[u].w, vv[u].w, ee[u].w, a);
            p4[i] = pp[u]; m4[i] = mm[u]; v4[i] = vv[u];
            if (p16) {  // bf16 shadow of the master weights: the GEMMs read it, no per-step cast kernels
                struct alignas(8) B4 { __hip_bfloat16 x, y, z, w; } o = {__float2bfloat16(pp[u].x), __float2bfloat16(pp[u].y),
                                                                         __float2bfloat16(pp[u].z), __float2bfloat16(pp[u].w)};
                reinterpret_cast<B4 *>(p16)[i] = o;
            }
            if (a.has_ema) e4[i] = ee[u];
            if (a.zero_grad) g4[i] = gg[u];
        }
    }
    // tail (n not a multiple of 4)
    const long t = (n4 << 2) + (long)blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x == 0 && t < n) {
        float ee = a.has_ema ? ema[t] : 0.0f;
        adam1(p[t], g[t], m[t], v[t], ee, a);
        if (p16) p16[t] = __float2bfloat16(p[t]);
        if (a.has_ema) ema[t] = ee;
    }
}

// ---- global gradient norm + clip coefficient (torch.nn.utils.clip_grad_norm_, norm_type 2; xqgan_train.py:456-458,471-473) ----------------
// Two launches, fixed summation order (deterministic): per-block sums of squares of g * grad_scale in double, then one block folds them:
// out[0] = total_norm, out[1] = min(1, max_norm / (total_norm + 1e-6)) — torch's clip_coef_clamped.  max_norm <= 0: out[1] = 1 (norm only).
constexpr int GN_BLOCK = 256;

__global__ __launch_bounds__(GN_BLOCK) void grad_sqsum_kernel(const float *__restrict__ g, long n, float scale, double *__restrict__ partials) {
    const long n4 = n >> 2;
    const long stride = (long)gridDim.x * GN_BLOCK;
    const float4 *g4 = reinterpret_cast<const float4 *>(g);
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    double acc = 0.0;
    int run = 0;
    for (long i = (long)blockIdx.x * GN_BLOCK + threadIdx.x; i < n4; i += stride) {
        float4 v = g4[i];
        v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
        acc0 = __builtin_fmaf(v.x, v.x, acc0); acc1 = __builtin_fmaf(v.y, v.y, acc1);
        acc2 = __builtin_fmaf(v.z, v.z, acc2); acc3 = __builtin_fmaf(v.w, v.w, acc3);
        if (++run == 64) {      // short fp32 runs folded into a double: the sum of 1.7e8 squares keeps ~1e-7 relative accuracy
            acc += (double)acc0 + (double)acc1 + (double)acc2 + (double)acc3;
            acc0 = acc1 = acc2 = acc3 = 0.f; run = 0;
        }
    }
    acc += (double)acc0 + (double)acc1 + (double)acc2 + (double)acc3;
    const long t = (n4 << 2) + threadIdx.x;
    if (blockIdx.x == 0 && t < n) { const double v = (double)(g[t] * scale); acc += v * v; }
    __shared__ double red[GN_BLOCK];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = GN_BLOCK / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partials[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(GN_BLOCK) void grad_norm_finalize_kernel(const double *__restrict__ partials, int nblocks, float max_norm,
                                                                       float *__restrict__ out) {
    __shared__ double red[GN_BLOCK];
    double acc = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += GN_BLOCK) acc += partials[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = GN_BLOCK / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float total = (float)sqrt(red[0]);
        out[0] = total;
        const float coef = max_norm > 0.0f ? max_norm / (total + 1e-6f) : 1.0f;
        out[1] = coef < 1.0f ? coef : 1.0f;      // a NaN / inf norm gives a NaN coefficient, as torch's (error_if_nonfinite = False)
        if (coef != coef) out[1] = coef;
    }
}

extern "C" size_t xq_grad_norm_workspace_bytes(void) { return (size_t)num_cus() * 8 * sizeof(double); }

extern "C" int xq_grad_norm_clip(const float *g, int64_t n, float grad_scale, float max_norm, void *workspace, size_t workspace_bytes,
                                 float *out2, xq_stream_t stream) {
    if (!out2 || !workspace || (n > 0 && !g)) return xq_set_error(XQ_EINVAL, "xq_grad_norm_clip: null pointer");
    if (n < 0) return xq_set_error(XQ_EINVAL, "%s: bad n (%ld)", "xq_grad_norm_clip", (long)n);
    if ((((uintptr_t)g) & 15) != 0 || (((uintptr_t)workspace) & 7) != 0) return xq_set_error(XQ_EINVAL, "xq_grad_norm_clip: g must be 16-byte, workspace 8-byte aligned");
    long blocks = (n / 4 + GN_BLOCK - 1) / GN_BLOCK;
    const long cap = (long)num_cus() * 8;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    if (workspace_bytes < (size_t)blocks * sizeof(double)) return xq_set_error(XQ_EINVAL, "xq_grad_norm_clip: workspace too small");
    hipLaunchKernelGGL(grad_sqsum_kernel, dim3((unsigned)blocks), dim3(GN_BLOCK), 0, (hipStream_t)stream, g, (long)n, grad_scale, (double *)workspace);
    hipLaunchKernelGGL(grad_norm_finalize_kernel, dim3(1), dim3(GN_BLOCK), 0, (hipStream_t)stream, (const double *)workspace, (int)blocks, max_norm, out2);
    return xq_check_launch("grad_norm_clip kernels");
}

static int adamw_launch(const char *fn, float *p, float *g, float *m, float *v, float *ema, void *p_bf16, int64_t n, float lr, float beta1,
                        float beta2, float eps, float weight_decay, int64_t step, const float *coeffs, float ema_decay, float grad_scale,
                        int zero_grad, xq_stream_t stream, const float *clip = nullptr) {
    if (n == 0) return XQ_OK;
    if (!p || !g || !m || !v) return xq_set_error(XQ_EINVAL, "%s: null pointer", fn);
    if (n < 0 || (!coeffs && step < 1)) return xq_set_error(XQ_EINVAL, "%s: bad n/step (%ld, %ld)", fn, (long)n, (long)step);
    if ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v | (uintptr_t)ema) & 15) != 0)
        return xq_set_error(XQ_EINVAL, "%s: arenas must be 16-byte aligned", fn);
    AdamArgs a;
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay;
    a.coeffs = coeffs;
    a.clip = clip;
    if (coeffs) { a.step_size = 0.0f; a.inv_bc2_sqrt = 0.0f; }
    else {
        const double bc1 = 1.0 - pow((double)beta1, (double)step);
        const double bc2 = 1.0 - pow((double)beta2, (double)step);
        a.step_size = (float)((double)lr / bc1);
        a.inv_bc2_sqrt = (float)(1.0 / sqrt(bc2));
    }
    a.ema_decay = ema_decay; a.grad_scale = grad_scale;
    a.zero_grad = zero_grad; a.has_ema = ema != nullptr;
    long blocks = (n / 4 + 255) / 256;
    const long cap = (long)num_cus() * 8;
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    const int pslot = xq::prof_begin(XQ_PROF_ADAMW, (double)n * (4.0 * ((ema ? 5 : 4) + (ema ? 4 : 3) + (zero_grad ? 1 : 0)) + (p_bf16 ? 2.0 : 0.0)), (hipStream_t)stream);
    hipLaunchKernelGGL(adamw_ema_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, ema, (__hip_bfloat16 *)p_bf16,
                       (long)n, a);
    xq::prof_end(pslot, (hipStream_t)stream);
    return xq_check_launch("adamw_ema_kernel");
}

extern "C" int xq_adamw_ema_step(float *p, float *g, float *m, float *v, float *ema, void *p_bf16, int64_t n, float lr,
                                 float beta1, float beta2, float eps, float weight_decay, int64_t step, float ema_decay,
                                 float grad_scale, int zero_grad, xq_stream_t stream) {
    return adamw_launch("xq_adamw_ema_step", p, g, m, v, ema, p_bf16, n, lr, beta1, beta2, eps, weight_decay, step, nullptr, ema_decay,
                        grad_scale, zero_grad, stream);
}

extern "C" int xq_adamw_ema_step_dev(float *p, float *g, float *m, float *v, float *ema, void *p_bf16, int64_t n, float lr,
                                     float beta1, float beta2, float eps, float weight_decay, const float *coeffs, float ema_decay,
                                     float grad_scale, int zero_grad, xq_stream_t stream) {
    if (!coeffs) return xq_set_error(XQ_EINVAL, "%s: null coefficient pointer", "xq_adamw_ema_step_dev");
    return adamw_launch("xq_adamw_ema_step_dev", p, g, m, v, ema, p_bf16, n, lr, beta1, beta2, eps, weight_decay, 0, coeffs, ema_decay,
                        grad_scale, zero_grad, stream);
}

// the general form: device-resident bias-correction factors (coeffs, nullable: then the 1-based `step` is used) and a device-resident clip
// coefficient (clip2 = the out2 of xq_grad_norm_clip, nullable)
extern "C" int xq_adamw_ema_step_ex(float *p, float *g, float *m, float *v, float *ema, void *p_bf16, int64_t n, float lr,
                                    float beta1, float beta2, float eps, float weight_decay, int64_t step, const float *coeffs,
                                    const float *clip2, float ema_decay, float grad_scale, int zero_grad, xq_stream_t stream) {
    return adamw_launch("xq_adamw_ema_step_ex", p, g, m, v, ema, p_bf16, n, lr, beta1, beta2, eps, weight_decay, step, coeffs, ema_decay,
                        grad_scale, zero_grad, stream, clip2);
}

// ---- transposed bf16 shadows of the Linear weights ------------------------------------------------------------------------------
// The data gradient of a Linear, g_x = g_y W, runs 7-10 % faster as an NT product on W^T ([in][out]: both operands K-major, staged by LDS-DMA
// and read with plain ds_read_b64) than as an NN product on W as stored (transpose reads) — profiles/r06_nn_vs_nt_transposed_weight.txt.
// One launch after the optimizer step rewrites every registered weight's transposed copy: `table` (device, int64 [n][5]) = {source offset,
// destination offset (elements from src / dst), rows, cols, first tile}, rows and cols multiples of 64, tiles numbered row-major inside a
// matrix.  HBM-bound, 4 B per element.
__global__ __launch_bounds__(256) void transpose_shadow_kernel(const unsigned short *__restrict__ src, unsigned short *__restrict__ dst,
                                                               const long *__restrict__ table, int n) {
    __shared__ unsigned short tile[64 * 66];        // pitch 33 words: the 8 rows a lane gathers and the 8 lanes of a column group hit distinct banks
    const long t = blockIdx.x;
    int lo = 0, hi = n - 1;
    while (lo < hi) {                               // last matrix whose first tile <= t
        const int mid = (lo + hi + 1) >> 1;
        if (table[5 * mid + 4] <= t) lo = mid; else hi = mid - 1;
    }
    const long *e = table + 5 * lo;
    const long rows = e[2], cols = e[3];
    const long tl = t - e[4], tcols = cols >> 6;
    const long r0 = (tl / tcols) << 6, c0 = (tl % tcols) << 6;
    const unsigned short *s = src + e[0] + r0 * cols + c0;
    unsigned short *d = dst + e[1] + c0 * rows + r0;
    const int lane8 = threadIdx.x & 7, grp = threadIdx.x >> 3;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int r = grp + 32 * p;
        const uint4 v = *reinterpret_cast<const uint4 *>(s + (long)r * cols + 8 * lane8);
        unsigned *w = reinterpret_cast<unsigned *>(tile + r * 66 + 8 * lane8);
        w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int c = grp + 32 * p;                 // source column = destination row
        unsigned o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            o[j] = (unsigned)tile[(8 * lane8 + 2 * j) * 66 + c] | ((unsigned)tile[(8 * lane8 + 2 * j + 1) * 66 + c] << 16);
        *reinterpret_cast<uint4 *>(d + (long)c * rows + 8 * lane8) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

extern "C" int xq_transpose_bf16_batched(const void *src, void *dst, const int64_t *table, int n_mats, int64_t tiles, xq_stream_t stream) {
    const char *fn = "xq_transpose_bf16_batched";
    if (n_mats < 0 || tiles < 0) return xq_set_error(XQ_EINVAL, "%s: negative count", fn);
    if (n_mats == 0 || tiles == 0) return XQ_OK;
    if (!src || !dst || !table) return xq_set_error(XQ_EINVAL, "%s: null pointer", fn);
    if (tiles > 0x7fffffffL) return xq_set_error(XQ_EINVAL, "%s: %ld tiles exceed the grid", fn, (long)tiles);
    static_assert(sizeof(long) == sizeof(int64_t), "table entries are read as long");
    hipLaunchKernelGGL(transpose_shadow_kernel, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, (const unsigned short *)src,
                       (unsigned short *)dst, (const long *)table, n_mats);
    return xq_check_launch("transpose_shadow_kernel");
}
