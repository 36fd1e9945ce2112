// xq_act.hpp — GELU value / derivative shared by the row kernels (xq_dense.hip) and the GEMM epilogues (xq_gemm.hip):
// nn.GELU() (exact, erf) of timm's Mlp (dino_enc/vision_transformer.py:295-339 via timm.layers.Mlp) and the tanh approximation
// of the DINO-S discriminator trunk.  One definition, so the fused and the stand-alone paths round identically.
//
// Round 4: everything works on PAIRS of values in packed fp32 (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32: two lanes' worth of
// arithmetic per VALU issue slot).  The fused fc1 / fc2-gradient GEMM epilogues spend ~11 k VALU cycles per 256 x 256 tile and SIMD
// on this function with the matrix pipe idle (profiles/r03_gemm_where_the_cycles_go.md: +38 % kernel time over the plain product);
// the packed form issues ~12 instead of ~20 VALU slots per element.
#pragma once
#include <hip/hip_runtime.h>

typedef float act_f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ act_f2 act_splat(float a) { return act_f2{a, a}; }
__device__ __forceinline__ act_f2 act_fma(act_f2 a, act_f2 b, act_f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ act_f2 act_rcp(act_f2 a) { return act_f2{__builtin_amdgcn_rcpf(a.x), __builtin_amdgcn_rcpf(a.y)}; }
__device__ __forceinline__ act_f2 act_exp2(act_f2 a) { return act_f2{__builtin_amdgcn_exp2f(a.x), __builtin_amdgcn_exp2f(a.y)}; }
__device__ __forceinline__ act_f2 act_abs(act_f2 a) { return act_f2{__builtin_fabsf(a.x), __builtin_fabsf(a.y)}; }

// erf through Abramowitz-Stegun 7.1.26 (|error| <= 1.5e-7, the size of erff's own fp32 rounding): one exp2 and one rcp per value,
// no branches; the exponential e = exp(-x^2/2) is the same one the GELU derivative needs.  (libm erff costs ~40 VALU ops
// and made the bf16 backward pass VALU-bound at 3.2 TB/s.)
// half_tail = 0.5 * (1 - erf(|x| / sqrt2)) = t * P(t) * e with the polynomial's coefficients pre-multiplied by 0.5 (exact: a power of two)
__device__ __forceinline__ void erf_half_tail2(act_f2 x, act_f2 ax, act_f2 &half_tail, act_f2 &e) {
    const act_f2 t = act_rcp(act_fma(ax, act_splat(0.3275911f * 0.70710678118654752440f), act_splat(1.0f)));
    e = act_exp2((x * x) * act_splat(-0.72134752044448170368f));   // exp(-x^2/2)
    act_f2 p = act_fma(t, act_splat(0.5f * 1.061405429f), act_splat(0.5f * -1.453152027f));
    p = act_fma(t, p, act_splat(0.5f * 1.421413741f));
    p = act_fma(t, p, act_splat(0.5f * -0.284496736f));
    p = act_fma(t, p, act_splat(0.5f * 0.254829592f));
    half_tail = (t * p) * e;
}

template <bool TANH> __device__ __forceinline__ act_f2 gelu_val2(act_f2 x) {
#ifdef XQ_ACT_DEBUG_CHEAP      // timing experiment only (profiles/r06_gelu_epilogue_cost.txt): what the fused epilogues cost WITHOUT the activation arithmetic
    return x * act_splat(0.5f);
#endif
    if (TANH) {   // F.gelu(approximate='tanh'): 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3))) = x * sigmoid(2u)
        const act_f2 w = act_fma((x * x) * act_splat(0.044715f), x, x);
        const act_f2 s = act_rcp(act_splat(1.0f) + act_exp2(w * act_splat(-2.8853900817779268f * 0.7978845608028654f)));
        return x * s;
    }
    // x * Phi(x) = max(x, 0) - |x| * half_tail   (x >= 0: x (1 - half_tail); x < 0: x half_tail)
    const act_f2 ax = act_abs(x);
    act_f2 ht, e;
    erf_half_tail2(x, ax, ht, e);
    const act_f2 relu = act_f2{fmaxf(x.x, 0.0f), fmaxf(x.y, 0.0f)};
    return act_fma(-ax, ht, relu);
}

template <bool TANH> __device__ __forceinline__ act_f2 gelu_grad2(act_f2 x) {
#ifdef XQ_ACT_DEBUG_CHEAP
    return act_splat(0.5f) + x * act_splat(0.125f);
#endif
    if (TANH) {
        const act_f2 x2 = x * x;
        const act_f2 w = act_fma(x2 * act_splat(0.044715f), x, x);
        const act_f2 sg = act_rcp(act_splat(1.0f) + act_exp2(w * act_splat(-2.8853900817779268f * 0.7978845608028654f)));   // 0.5 (1 + tanh u)
        const act_f2 du = act_fma(x2, act_splat(0.134145f * 0.7978845608028654f), act_splat(0.7978845608028654f));
        // sg + x * (1 - tanh^2 u)/2 * du,  (1 - tanh^2)/2 = 2 sg (1 - sg)
        const act_f2 k = ((x + x) * sg) * (act_splat(1.0f) - sg);
        return act_fma(k, du, sg);
    }
    // Phi(x) + x phi(x):  Phi = 0.5 + copysign(0.5 - half_tail, x)
    const act_f2 ax = act_abs(x);
    act_f2 ht, e;
    erf_half_tail2(x, ax, ht, e);
    const act_f2 s = act_splat(0.5f) - ht;
    const act_f2 cdf = act_splat(0.5f) + act_f2{__builtin_copysignf(s.x, x.x), __builtin_copysignf(s.y, x.y)};
    return act_fma(x * act_splat(0.39894228040143267794f), e, cdf);
}

// single values: the pair functions on a duplicated argument (same roundings as every fused path)
template <bool TANH> __device__ __forceinline__ float gelu_val(float x) { return gelu_val2<TANH>(act_splat(x)).x; }
template <bool TANH> __device__ __forceinline__ float gelu_grad(float x) { return gelu_grad2<TANH>(act_splat(x)).x; }
