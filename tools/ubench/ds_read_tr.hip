// Probe of the gfx950 LDS transpose read (ds_read_b64_tr_b16): which 16-bit elements does each lane receive?
// LDS holds its own element index; every lane passes an address; the 4 returned shorts are printed per lane.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void probe(int stride_shorts, short *out) {
    __shared__ short lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
    __syncthreads();
    s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((s4 __attribute__((address_space(3))) *)(lds + threadIdx.x * stride_shorts));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}
int main() {
    short *d; hipMalloc(&d, 64 * 4 * 2);
    short h[256];
    for (int stride : {4, 64, 72}) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, stride, d);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("stride %d shorts per lane\n", stride);
        for (int l = 0; l < 64; ++l) {
            printf("lane %2d:", l);
            for (int j = 0; j < 4; ++j) printf(" %5d(l%d+%d)", h[l * 4 + j], h[l * 4 + j] / stride, h[l * 4 + j] % stride);
            printf("\n");
        }
    }
    return 0;
}
