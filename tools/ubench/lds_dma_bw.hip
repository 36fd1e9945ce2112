// Microbenchmark: L2 -> CU throughput of (A) global_load_lds_dwordx4 (LDS-DMA), (B) global_load_dwordx4 into registers,
// (C) B + ds_write_b128, per CU, with the source resident in L2 (each block re-reads its own 128 KiB window) —
// contiguous 1 KiB per wave instruction or 128-byte rows at a 4608-byte pitch (the packed-qkv pattern).
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/lds_dma_bw.hip -o tools/ubench/lds_dma_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) void gbl_void;
constexpr int ITERS = 128;         // outer repeats over the window
constexpr int WIN = 64 * 1024;     // bytes per block window

template <int MODE, int STRIDED, int DEPTH>
__global__ __launch_bounds__(256) void k(const char *src, float *sink, long win_stride) {
    __shared__ __attribute__((aligned(1024))) char smem[DEPTH * 4096 > 32768 ? DEPTH * 4096 : 32768];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const char *win = src + (long)(blockIdx.x & 255) * win_stride;   // 16 MiB in total: 2 MiB per XCD, L2-resident
    // wave instruction j of this wave covers bytes [(4 j + wave) 1024, +1024) of the window (contiguous) or rows 8 (4 j + wave) .. +7
    unsigned lane_off = STRIDED ? (unsigned)((lane >> 3) * 4608 + (lane & 7) * 16) : (unsigned)(lane * 16);
    const unsigned step = STRIDED ? 8 * 4608 : 1024;
    uint4 acc = make_uint4(0, 0, 0, 0);
    for (int it = 0; it < ITERS; ++it) {
        for (int j0 = 0; j0 < WIN / 4096; j0 += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const char *p = win + (long)(4 * (j0 + d) + wave) * step + lane_off;
                if (MODE == 0) {
                    __builtin_amdgcn_global_load_lds((gbl_void *)p, (lds_void *)(smem + d * 4096 + wave * 1024), 16, 0, 0);
                } else {
                    const uint4 v = *reinterpret_cast<const uint4 *>(p);
                    if (MODE == 2) *reinterpret_cast<uint4 *>(smem + d * 4096 + wave * 1024 + lane * 16) = v;
                    else { acc.x ^= v.x; acc.y ^= v.y; acc.z ^= v.z; acc.w ^= v.w; }
                }
            }
            if (MODE == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    }
    __syncthreads();
    if (MODE != 1) acc = *reinterpret_cast<const uint4 *>(smem + tid * 16);
    if (acc.x == 0x12345678u) sink[tid] = 1.0f;
}

template <int MODE, int STRIDED, int DEPTH> static void run(const char *src, float *sink, int bpc, const char *name) {
    const int blocks = 256 * bpc;
    const long win_stride = STRIDED ? (long)WIN / 128 * 4608 : WIN;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, STRIDED, DEPTH>), dim3(blocks), dim3(256), 0, 0, src, sink, win_stride);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL((k<MODE, STRIDED, DEPTH>), dim3(blocks), dim3(256), 0, 0, src, sink, win_stride);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)blocks * ITERS * WIN;
    printf("%-34s %s depth %2d, %d block(s)/CU: %8.1f us  %7.2f TB/s  %6.1f GB/s/CU  %5.1f B/clk/CU @2.4GHz\n", name, STRIDED ? "strided" : "contig ",
           DEPTH, bpc, ms * 1e3, bytes / ms / 1e9, bytes / ms / 1e6 / 256, bytes / ms / 1e6 / 256 / 2.4);
}

int main() {
    char *src; float *sink;
    const size_t bytes = (size_t)256 * 4 * (WIN / 128) * 4608 + (1 << 20);
    hipMalloc(&src, bytes); hipMemset(src, 1, bytes); hipMalloc(&sink, 4096);
    for (int bpc : {1, 2, 4}) {
        if (bpc == 1) { run<0, 0, 8>(src, sink, 1, "A LDS-DMA"); run<0, 1, 8>(src, sink, 1, "A LDS-DMA"); run<1, 0, 8>(src, sink, 1, "B load -> VGPR"); run<1, 1, 8>(src, sink, 1, "B load -> VGPR"); run<2, 0, 8>(src, sink, 1, "C load -> VGPR -> ds_write"); run<2, 1, 8>(src, sink, 1, "C load -> VGPR -> ds_write"); run<0, 1, 4>(src, sink, 1, "A LDS-DMA"); run<0,1,2>(src, sink, 1, "A LDS-DMA"); }
        if (bpc == 2) { run<0, 0, 8>(src, sink, 2, "A LDS-DMA"); run<0, 1, 8>(src, sink, 2, "A LDS-DMA"); run<1, 0, 8>(src, sink, 2, "B load -> VGPR"); run<1, 1, 8>(src, sink, 2, "B load -> VGPR"); run<2, 0, 8>(src, sink, 2, "C load -> VGPR -> ds_write"); run<2, 1, 8>(src, sink, 2, "C load -> VGPR -> ds_write"); run<0, 1, 4>(src, sink, 2, "A LDS-DMA"); }
        if (bpc == 4) { run<0, 0, 8>(src, sink, 4, "A LDS-DMA"); run<0, 1, 8>(src, sink, 4, "A LDS-DMA"); run<1, 0, 8>(src, sink, 4, "B load -> VGPR"); run<1, 1, 8>(src, sink, 4, "B load -> VGPR"); run<2, 1, 8>(src, sink, 4, "C load -> VGPR -> ds_write"); }
    }
    return 0;
}
