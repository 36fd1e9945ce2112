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
    const long n4 = n >> 2;
    const long stride = (long)gridDim.x * 256;
    float4 *p4 = reinterpret_cast<float4 *>(p), *g4 = reinterpret_cast<float4 *>(g), *m4 = reinterpret_cast<float4 *>(m),
           *v4 = reinterpret_cast<float4 *>(v), *e4 = reinterpret_cast<float4 *>(ema);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        float4 pp = p4[i], gg = g4[i], mm = m4[i], vv = v4[i];
        float4 ee = a.has_ema ? e4[i] : make_float4(0, 0, 0, 0);
        adam1(pp.x, gg.x, mm.x, vv.x, ee.x, a);
        adam1(pp.y, gg.y, mm.y, vv.y, ee.y, a);
        adam1(pp.z, gg.z, mm.z, vv.z, ee.z, a);
        adam1(pp.w, gg.w, mm.w, vv.w, ee.w, a);
        p4[i] = pp; m4[i] = mm; v4[i] = vv;
        if (p16) {  // bf16 shadow of the master weights: the GEMMs read it, no per-step cast kernels
            struct alignas(8) B4 { __hip_bfloat16 x, y, z, w; } o = {__float2bfloat16(pp.x), __float2bfloat16(pp.y),
                                                                     __float2bfloat16(pp.z), __float2bfloat16(pp.w)};
            reinterpret_cast<B4 *>(p16)[i] = o;
        }
        if (a.has_ema) e4[i] = ee;
        if (a.zero_grad) g4[i] = gg;
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

static int adamw_launch(const char *fn, float *p, float *g, float *m, float *v, float *ema, void *p_bf16, int64_t n, float lr, float beta1,
                        float beta2, float eps, float weight_decay, int64_t step, const float *coeffs, float ema_decay, float grad_scale,
                        int zero_grad, xq_stream_t stream) {
    if (n == 0) return XQ_OK;
    if (!p || !g || !m || !v) return xq_set_error(XQ_EINVAL, "%s: null pointer", fn);
    if (n < 0 || (!coeffs && step < 1)) return xq_set_error(XQ_EINVAL, "%s: bad n/step (%ld, %ld)", fn, (long)n, (long)step);
    if ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v | (uintptr_t)ema) & 15) != 0)
        return xq_set_error(XQ_EINVAL, "%s: arenas must be 16-byte aligned", fn);
    AdamArgs a;
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.weight_decay = weight_decay;
    a.coeffs = coeffs;
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
    hipLaunchKernelGGL(adamw_ema_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, ema, (__hip_bfloat16 *)p_bf16,
                       (long)n, a);
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
