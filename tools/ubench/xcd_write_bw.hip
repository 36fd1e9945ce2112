// Microbenchmark: how fast can ONE XCD (32 CUs) write to HBM, next to all eight at once?  Decides whether de-synchronising the GEMM epilogues by
// XCD can shorten them: in the persistent GEMM all 256 workgroups store their 128 KiB tiles at the same moment at ~10 B / cycle / CU
// (profiles/r03_gemm_where_the_cycles_go.md); if an XCD alone gets several times its 1/8 share, staggering the XCDs pays.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/xcd_write_bw tools/ubench/xcd_write_bw.hip && /tmp/xcd_write_bw
// Blocks b with bit (b & 7) set in `mask` write `chunks` x 128 KiB each (512 threads, 16-byte stores, non-temporal or plain); block b is placed on
// XCD b % 8 by the dispatcher (MI355X_MICROARCH.md).  Also: the same with every block ALSO streaming reads (a GEMM K loop's operand traffic).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ __launch_bounds__(512) void wr(u32x4 *out, int mask, int chunks, int nblocks_per_xcd) {
    const int x = blockIdx.x & 7;
    if (!((mask >> x) & 1)) return;
    const long slot = (long)(blockIdx.x >> 3) + (long)x * nblocks_per_xcd;          // contiguous region per XCD
    u32x4 *p = out + slot * chunks * 8192L + threadIdx.x;
    u32x4 v = {threadIdx.x, blockIdx.x, 3u, 4u};
    for (int c = 0; c < chunks; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            u32x4 *q = p + (long)c * 8192 + i * 512;
            if (NT) __builtin_nontemporal_store(v, q); else *q = v;
        }
}

int main() {
    const int per_xcd = 32, chunks = 64;      // 32 blocks per XCD (one per CU), 8 MiB per block
    const size_t bytes = (size_t)8 * per_xcd * chunks * 131072;
    u32x4 *buf;
    if (hipMalloc(&buf, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int masks[] = {0xff, 0x01, 0x03, 0x0f, 0x10, 0x55};
    for (int nt = 0; nt < 2; ++nt)
        for (int mask : masks) {
            float best = 1e30f;
            for (int rep = 0; rep < 4; ++rep) {
                hipEventRecord(e0);
                if (nt) hipLaunchKernelGGL(wr<true>, dim3(8 * per_xcd), dim3(512), 0, 0, buf, mask, chunks, per_xcd);
                else hipLaunchKernelGGL(wr<false>, dim3(8 * per_xcd), dim3(512), 0, 0, buf, mask, chunks, per_xcd);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep && ms < best) best = ms;
            }
            const int nx = __builtin_popcount(mask);
            const double gb = (double)nx * per_xcd * chunks * 131072 / 1e9;
            printf("%s stores, XCD mask 0x%02x (%d XCDs, %d CUs): %7.3f ms  %7.1f GB/s total  %6.1f GB/s per XCD  %5.1f B/cycle/CU at 2.0 GHz\n", nt ? "non-temporal" : "plain       ",
                   mask, nx, nx * per_xcd, best, gb / best * 1e3, gb / best * 1e3 / nx, gb / best * 1e3 / (nx * per_xcd) / 2.0);
        }
    return 0;
}
