// xq_act.hpp — GELU value / derivative shared by the row kernels (xq_dense.hip) and the GEMM epilogues (xq_gemm.hip):
// nn.GELU() (exact, erf) of timm's Mlp (dino_enc/vision_transformer.py:295-339 via timm.layers.Mlp) and the tanh approximation
// of the DINO-S discriminator trunk.  One definition, so the fused and the stand-alone paths round identically.
#pragma once
#include <hip/hip_runtime.h>

// erf through Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, the size of erff's own fp32 rounding): one exp2 and one rcp,
// no branches; the exponential e = exp(-x^2/2) is the same one the GELU derivative needs.  (libm erff costs ~40 VALU ops
// and made the bf16 backward pass VALU-bound at 3.2 TB/s.)
__device__ __forceinline__ void erf_cdf(float x, float &cdf, float &e) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
    e = __builtin_amdgcn_exp2f(x * x * -0.72134752044448170368f);   // exp(-x^2/2)
    float p = fmaf(t, 1.061405429f, -1.453152027f);
    p = fmaf(t, p, 1.421413741f);
    p = fmaf(t, p, -0.284496736f);
    p = fmaf(t, p, 0.254829592f);
    const float half_tail = 0.5f * t * p * e;                        // 0.5 * (1 - erf(|x|/sqrt2))
    cdf = x >= 0.0f ? 1.0f - half_tail : half_tail;
}
template <bool TANH> __device__ __forceinline__ float gelu_val(float x) {
    if (TANH) {   // F.gelu(approximate='tanh'): 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3))) = x * sigmoid(2u)
        const float u = 0.7978845608028654f * fmaf(0.044715f * x * x, x, x);
        return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.8853900817779268f * u));
    }
    float cdf, e;
    erf_cdf(x, cdf, e);
    return x * cdf;
}
template <bool TANH> __device__ __forceinline__ float gelu_grad(float x) {
    if (TANH) {
        const float x2 = x * x;
        const float u = 0.7978845608028654f * fmaf(0.044715f * x2, x, x);
        const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.8853900817779268f * u));   // 0.5 (1 + tanh u)
        const float du = 0.7978845608028654f * fmaf(0.134145f, x2, 1.0f);
        return fmaf(2.0f * x * sg * (1.0f - sg), du, sg);   // sg + x * (1 - tanh^2 u)/2 * du,  (1 - tanh^2)/2 = 2 sg (1 - sg)
    }
    float cdf, e;
    erf_cdf(x, cdf, e);
    return fmaf(x * 0.39894228040143267794f, e, cdf);
}

