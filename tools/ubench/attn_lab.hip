// Lab harness for the attention kernels of csrc/xq_attn.hip: runs them outside Python on random packed-qkv data, checks the
// streaming forward against the round-1 forward and times every kernel with HIP events.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 tools/ubench/attn_lab.hip -o tools/ubench/attn_lab
//   tools/ubench/attn_lab [B N H iters]
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
#include <string.h>
#include <vector>
#include "../../imagefolder_amd/csrc/xq_attn.hip"
#include "attn_fwd2_abl.inc"

thread_local char g_err[512];
int xq_set_error(int code, const char *fmt, const char *a, long b, long c) { fprintf(stderr, fmt, a, b, c); fprintf(stderr, "\n"); return code; }
int xq_check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", what, hipGetErrorString(e)); return 1; }
    return 0;
}
namespace xq {
int prof_begin(int, double, hipStream_t) { return -1; }
void prof_end(int, hipStream_t) {}
}

static unsigned short f2bf(float f) {
    unsigned u; memcpy(&u, &f, 4);
    u += 0x7fff + ((u >> 16) & 1);
    return (unsigned short)(u >> 16);
}
static float bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

template <class F> static float time_ms(F fn, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) fn();
    hipEventRecord(e0, 0);
    for (int i = 0; i < iters; ++i) fn();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / iters;
}

int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 128, N = argc > 2 ? atoi(argv[2]) : 513, H = argc > 3 ? atoi(argv[3]) : 12;
    const int iters = argc > 4 ? atoi(argv[4]) : 20;
    const long nq = (long)B * N * 3 * H * 64, no = (long)B * N * H * 64, nl = (long)B * H * N;
    std::vector<unsigned short> hq(nq);
    unsigned long long st = 88172645463325252ull;
    for (long i = 0; i < nq; ++i) {   // sum of 4 uniforms, unit variance
        float a = 0;
        for (int j = 0; j < 4; ++j) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; a += (float)((st >> 40) & 0xffff) / 65536.0f - 0.5f; }
        hq[i] = f2bf(a * 1.7320508f);
    }
    short *qkv, *o1, *o2; float *l1, *l2;
    hipMalloc(&qkv, nq * 2); hipMalloc(&o1, no * 2); hipMalloc(&o2, no * 2); hipMalloc(&l1, nl * 4); hipMalloc(&l2, nl * 4);
    hipMemcpy(qkv, hq.data(), nq * 2, hipMemcpyHostToDevice);
    hipMemset(o1, 0, no * 2); hipMemset(o2, 0, no * 2);
    const float scale = 0.125f, c = scale * 1.4426950408889634f;
    const int nqb = (N + 127) / 128, G8 = (B * H + 7) / 8 * 8;
    auto v0 = [&] { hipLaunchKernelGGL(attn_fwd_kernel, dim3(G8 * nqb), dim3(256), 0, 0, qkv, B, N, H, c, o1, l1, nqb); };
    auto v2 = [&] { hipLaunchKernelGGL(attn_fwd2_abl_kernel<0>, dim3(G8 * nqb), dim3(256), 0, 0, qkv, B, N, H, c, o2, l2, nqb); };
    v0(); v2();
    if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
    std::vector<unsigned short> h1(no), h2(no);
    std::vector<float> hl1(nl), hl2(nl);
    hipMemcpy(h1.data(), o1, no * 2, hipMemcpyDeviceToHost); hipMemcpy(h2.data(), o2, no * 2, hipMemcpyDeviceToHost);
    hipMemcpy(hl1.data(), l1, nl * 4, hipMemcpyDeviceToHost); hipMemcpy(hl2.data(), l2, nl * 4, hipMemcpyDeviceToHost);
    double md = 0, ml = 0; long bad = 0;
    for (long i = 0; i < no; ++i) {
        const double d = fabs((double)bf2f(h1[i]) - (double)bf2f(h2[i]));
        if (!(d <= 0.02)) { if (bad < 5) printf("  out[%ld] (row %ld col %ld): v0 %g v2 %g\n", i, i / (H * 64), i % (H * 64), bf2f(h1[i]), bf2f(h2[i])); ++bad; }
        if (d > md) md = d;
    }
    for (long i = 0; i < nl; ++i) { const double d = fabs((double)hl1[i] - (double)hl2[i]); if (!(d <= ml)) ml = d > ml ? d : (d != d ? 1e30 : ml); }
    printf("B=%d N=%d H=%d: fwd2 vs fwd: max |dO| %.4g (%ld above 0.02), max |dlse| %.4g\n", B, N, H, md, bad, ml);
    const double fl = 4.0 * B * H * (double)N * N * 64.0;
    const float t0 = time_ms(v0, iters), t2 = time_ms(v2, iters);
    printf("  fwd  (round 1) %8.1f us  %7.1f TF/s\n  fwd2 (stream)  %8.1f us  %7.1f TF/s\n", t0 * 1e3, fl / t0 / 1e9, t2 * 1e3, fl / t2 / 1e9);
#define ABL_RUN(A, WHAT) { auto f = [&] { hipLaunchKernelGGL(attn_fwd2_abl_kernel<A>, dim3(G8 * nqb), dim3(256), 0, 0, qkv, B, N, H, c, o2, l2, nqb); }; \
        const float t = time_ms(f, iters); printf("  ablation %3d %-44s %8.1f us\n", A, WHAT, t * 1e3); }
    if (argc > 5) {
        ABL_RUN(2048, "last key in a tile of its own (no v_dot2 prologue)")
        ABL_RUN(128, "no vmcnt wait in the loop (wrong results)")
        ABL_RUN(256, "no barrier in the loop (wrong results)")
        ABL_RUN(384, "no vmcnt wait, no barrier")
        ABL_RUN(512, "K/V of 8 (b, h) only (L2-hot)")
        ABL_RUN(79, "data movement only (Q, K/V DMA, barriers)")
        ABL_RUN(1, "no exp2 (scaled copy)")
        ABL_RUN(65, "no exp2, no max chain")
        ABL_RUN(32, "no ones-MFMA row sums")
        ABL_RUN(2, "no PV / row-sum MFMAs, no V reads")
        ABL_RUN(4, "no S MFMAs, no K reads")
        ABL_RUN(6, "no MFMAs at all")
        ABL_RUN(8, "no output store")
        ABL_RUN(16, "no K/V DMA")
        ABL_RUN(24, "no DMA, no store")
        ABL_RUN(71, "no MFMA, no exp, no max")
        ABL_RUN(95, "nothing but Q load + barriers")
    }
    return bad ? 2 : 0;
}
