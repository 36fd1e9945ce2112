// Microbenchmark: do v_mfma_f32_32x32x2_f32 and fp32 VALU work overlap on one SIMD of gfx950?
//   A: MFMA only (dependent chain)         B: VALU only (independent v_fma)        C: same wave, interleaved
//   D: two waves per SIMD, one MFMA-only and one VALU-only (wave-uniform role)
// Prints cycles per iteration (s_memtime) for each; overlap => C,D ~ max(A,B); sharing => ~A+B.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int ITERS = 2000;
constexpr int NV = 32;  // VALU fma per MFMA

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, int nwaves_active) {
    const int wave = threadIdx.x >> 6;
    f32x16 acc; for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    float v[8]; for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i;
    float a = threadIdx.x * 1e-4f, b = 1.0001f;
    bool do_mfma = (MODE == 0) || (MODE == 2) || (MODE == 3 && wave < 4);
    bool do_valu = (MODE == 1) || (MODE == 2) || (MODE == 3 && wave >= 4);
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; ++it) {
        if (do_mfma) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
        if (do_valu) {
#pragma unroll
            for (int j = 0; j < NV; ++j) v[j & 7] = __builtin_fmaf(v[j & 7], b, a);
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0; for (int i = 0; i < 16; ++i) s += acc[i]; for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    unsigned long long h[256 * 8];
    const char* names[] = {"A mfma only (8 waves/CU)", "B valu only (8 waves/CU)", "C interleaved same wave (8 waves/CU)", "D waves0-3 mfma, waves4-7 valu"};
    for (int threads : {256, 512}) {
        printf("block = %d threads (%d wave(s) per SIMD)\n", threads, threads / 256);
        for (int mode = 0; mode < 4; ++mode) {
            if (mode == 3 && threads == 256) continue;
            for (int rep = 0; rep < 2; ++rep) {
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(threads), 0, 0, out, cyc, 0);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(threads), 0, 0, out, cyc, 0);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(threads), 0, 0, out, cyc, 0);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(threads), 0, 0, out, cyc, 0);
                hipDeviceSynchronize();
            }
            hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
            double lo = 0, hi = 0; int nw = threads / 64;
            for (int b = 0; b < 256; ++b) for (int w = 0; w < nw; ++w) { if (w < 4) lo += h[b * 8 + w]; else hi += h[b * 8 + w]; }
            printf("  %-40s cycles/iter: waves0-3 %.1f", names[mode], lo / (256.0 * 4) / ITERS);
            if (nw > 4) printf("  waves4-7 %.1f", hi / (256.0 * 4) / ITERS);
            printf("   (1 MFMA + %d v_fma per iter)\n", NV);
        }
    }
    return 0;
}
