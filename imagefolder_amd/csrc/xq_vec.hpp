// xq_vec.hpp — activation-dtype helpers shared by the row/column kernels: bf16 <-> fp32 and 16-byte vector loads/stores.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_bf16.h>

typedef __hip_bfloat16 bf16;

template <typename T> __device__ __forceinline__ float to_f(T v);
template <> __device__ __forceinline__ float to_f<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f<bf16>(bf16 v) { return __bfloat162float(v); }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f<bf16>(float v) { return __float2bfloat16(v); }

// vector of VEC elements of T, loaded/stored in one instruction
template <typename T, int VEC> struct Pack { T v[VEC]; };

template <typename T, int VEC>
__device__ __forceinline__ void load_vec(const T *p, float (&out)[VEC]) {
    const Pack<T, VEC> pk = *reinterpret_cast<const Pack<T, VEC> *>(p);
#pragma unroll
    for (int j = 0; j < VEC; ++j) out[j] = to_f<T>(pk.v[j]);
}
template <typename T, int VEC>
__device__ __forceinline__ void store_vec(T *p, const float (&in)[VEC]) {
    Pack<T, VEC> pk;
#pragma unroll
    for (int j = 0; j < VEC; ++j) pk.v[j] = from_f<T>(in[j]);
    *reinterpret_cast<Pack<T, VEC> *>(p) = pk;
}

