// xq_common.hpp — shared helpers for the gfx950 quantizer kernels (device + host side).
//
// ARITHMETIC CONTRACT (identical to oracle/xq_oracle.c, which is the parity checker):
//   (A1) channel-axis dot products / sums of squares are sequential fp32 fmaf chains, ascending
//        channel, starting from +0.0f — exactly what a v_mfma_f32_32x32x2_f32 K-loop computes;
//   (A2) l2-normalise(x) = x / max(sqrtf(chain(x,x)), 1e-12f) with IEEE-rounded sqrt and division
//        (built with -fhip-fp32-correctly-rounded-divide-sqrt);
//   (A3) d = fl(fl(|z|^2 + |e|^2) - 2*dot);  (A4) lowest index wins ties.
// The translation unit is compiled with -ffp-contract=off so only explicit fmaf() fuses.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define XQ_EPS 1e-12f

namespace xq {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// float -> uint32 whose unsigned order equals the float order (NaN sorts last).
__device__ __forceinline__ uint32_t f2ord(float f) {
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(uint32_t u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

// out_mask of the convolutions (the ReLU of the layer BELOW a data gradient: aten::threshold_backward(g, y, 0) folded into the store of g):
// of a packed bf16 pair of the output each half survives where the mask's bf16 value is > 0 — read as a signed 16-bit integer:
// positive and not zero (the mask is a ReLU output: no NaN convention needed beyond "NaN with the sign clear passes", as ATen's y > 0 ... does not;
// a NaN activation has poisoned the step long before)
__device__ __forceinline__ uint32_t keep_where_positive(uint32_t v, uint32_t m) {
    const uint32_t lo = (int16_t)(m & 0xffffu) > 0 ? 0x0000ffffu : 0u;
    const uint32_t hi = (int32_t)m >= 0x00010000 ? 0xffff0000u : 0u;
    return v & (lo | hi);
}

// (A1)+(A2) on a register-resident row; returns the clamped norm.
template <int C>
__device__ __forceinline__ float l2norm_row(const float (&x)[C], float (&y)[C]) {
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < C; ++k) s = __builtin_fmaf(x[k], x[k], s);
    float n = __builtin_sqrtf(s);
    n = (n > XQ_EPS) ? n : XQ_EPS;
#pragma unroll
    for (int k = 0; k < C; ++k) y[k] = x[k] / n;
    return n;
}

template <int C>
__device__ __forceinline__ float chain_sq(const float (&x)[C]) {
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < C; ++k) s = __builtin_fmaf(x[k], x[k], s);
    return s;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// nearest-code search modes (same values as include/xq_ops.h)
enum { MODE_L2_NORMED = 0, MODE_L2_RAW = 1, MODE_COSINE = 2 };

// prologue shared by the assign kernels: loads this lane's token (both wave halves hold the same 32 tokens),
// normalises it (A2), leaves the MFMA A fragments in a[] and |zhat|^2 per accumulator row in zzr[].
template <int C, int MODE>
__device__ __forceinline__ void load_tokens(const float *__restrict__ z, long N, int HW, long tok0, int lane,
                                            float (&a)[C / 2], float (&zzr)[16]) {
    const int h = lane >> 5, li = lane & 31;
    long n = tok0 + li;
    if (n > N - 1) n = N - 1;
    const long b = n / HW;
    const int hw = (int)(n - b * HW);
    const float *base = z + (size_t)b * C * HW + hw;
    // two streaming passes keep register pressure at C/2: pass 1 = norm chain, pass 2 = zhat, |zhat|^2 chain
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < C; ++k) { const float x = base[(size_t)k * HW]; s = __builtin_fmaf(x, x, s); }
    float nrm = __builtin_sqrtf(s);
    nrm = (nrm > XQ_EPS) ? nrm : XQ_EPS;
    float zz = 0.0f;
#pragma unroll
    for (int k = 0; k < C; ++k) {
        const float x = base[(size_t)k * HW];
        const float zh = (MODE == MODE_L2_RAW) ? x : x / nrm;
        zz = __builtin_fmaf(zh, zh, zz);
        if ((k & 1) == 0) a[k >> 1] = zh;                // even channel: kept by the lower half
        else a[k >> 1] = h ? zh : a[k >> 1];             // odd channel: kept by the upper half
    }
    if (MODE == MODE_COSINE) zz = 0.0f;
    // accumulator register r of this lane is token row (r&3) + 8*(r>>2) + 4*h of the tile
#pragma unroll
    for (int r = 0; r < 16; ++r) zzr[r] = __shfl(zz, (r & 3) + 8 * (r >> 2) + 4 * h);
}

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

}  // namespace xq
