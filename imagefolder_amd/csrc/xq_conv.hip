// xq_conv.hip — 3x3 convolution (stride 1, pad 1) as an implicit GEMM on bf16 MFMA, NHWC activations (gfx950).
//
// Replaces the cuDNN/MIOpen conv3x3 calls of the reference's CNN encoder/decoder (tokenizer/tokenizer_image/
// xqgan_model.py:454-622: conv_in/ResnetBlock.conv1/conv2/Upsample.conv/conv_out — 99 % of its 391 GFLOP/img) and of the
// LPIPS VGG16 trunk (lpips.py:118-155).  On gfx950 MIOpen serves these shapes with igemm_*_nhwc_bf16 kernels at
// ~110 TFLOP/s (profiles/r01_train_step_full_kernel_stats.txt).
//
//   Y[b,y,x,n] = act( bias[n] + sum_{ky,kx,c} X[b, y+ky-1, x+kx-1, c] * W[n][c][ky][kx] )
// GEMM view: M = B*H*W output pixels, N = Cout, K = 9*Cin with k = (ky*3+kx)*Cin + c (channel fastest: every K-slice of
// 32 is a contiguous 64-byte run of one input pixel -> 16-byte coalesced gathers straight from NHWC, no im2col buffer).
// Weights are pre-packed once per update to Wp[n][k] (bf16), the same K order.
// The data gradient is the same kernel run on dY with Wp'[c][(2-ky)*3+(2-kx)][n] (rotated taps, swapped channels).
//
// Tile: 128 pixels x BN channels x 64 K per step (one tap's 64-channel run), 256 threads = 2x2 waves, each wave 64 x BN/2
// through v_mfma_f32_32x32x16_bf16 (fp32 accumulate); A/B tiles staged global -> VGPR -> LDS (row pitch 144 B:
// conflict-free ds_read_b128 fragments), double-buffered, one barrier per K step (16 or 8 MFMAs per wave between barriers).  Epilogue: + bias, optional ReLU, bf16 NHWC store.
#include "xq_common.hpp"
#include "xq_internal.hpp"
#include "../../include/xq_ops.h"

#include <hip/hip_bf16.h>
#include <cstdlib>
#include "xq_vec.hpp"

using namespace xq;

typedef short bf16x8 __attribute__((ext_vector_type(8)));   // 8 bf16 = one MFMA A/B fragment (4 VGPRs)

static constexpr int CV_BM = 128;

// 16-byte load whose address is always valid (index clamped by the caller) and whose value is zeroed afterwards: a
// "cond ? *p : zero" select makes hipcc pick between a global and a private pointer and issue flat loads through scratch
__device__ __forceinline__ uint4 cv_load16_or_zero(const __hip_bfloat16 *p, bool ok) {
    uint4 v = *reinterpret_cast<const uint4 *>(p);
    if (!ok) v = make_uint4(0u, 0u, 0u, 0u);
    return v;
}

// BK = K elements per step: 64 for the 128-channel tile (16 MFMAs per wave between barriers), 32 for the 64-channel tile
// (small-K layers are latency-bound: the 30 KB footprint keeps 5 blocks per CU in flight instead of 2)
template <int BN, int BK, bool RELU>
__global__ __launch_bounds__(256) void conv3x3_kernel(const __hip_bfloat16 *__restrict__ X, const __hip_bfloat16 *__restrict__ Wp,
                                                      const float *__restrict__ bias, long M, int H, int Wd, int Cin, int Cout,
                                                      __hip_bfloat16 *__restrict__ Y) {
    constexpr int WN = BN / 2;          // channels per wave
    constexpr int NT = WN / 32;         // 32-wide N tiles per wave (2 for BN=128, 1 for BN=64)
    // LDS rows: BK = 64 -> 144-byte pitch (padded: conflict-free 16-byte stores and ds_read_b128 fragment groups);
    // BK = 32 -> unpadded 64-byte rows with the 16-byte chunk index XOR-ed by (row >> 2) & 3: with any padded pitch the two
    // 64-byte rows of one 8-lane store group overlap by 4 banks (33 % of the LDS cycles were conflicts, profiles/r01_conv3x3_pmc.txt);
    // the swizzle makes both the stores and the fragment-read lane groups {0-3,12-15,20-27}, ... hit 16 distinct bank quads
    constexpr bool SWZ = (BK == 32);
    constexpr int CV_BK = BK, CV_PITCH = SWZ ? BK : BK + 8;
    constexpr int CPT = BK / 32;        // 16-byte chunks per thread and row
    __shared__ __attribute__((aligned(16))) short lds[2][(CV_BM + BN) * CV_PITCH];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // XCD-aware tile order (workgroups are dispatched round-robin over the 8 XCDs, each with its own L2): XCD k walks a
    // contiguous range of pixel tiles, channel tiles of one pixel tile back to back, so that the 3x3 halo rows and the
    // A tile shared by the Cout/BN channel blocks are re-read from that XCD's L2 instead of HBM (PMC before the remap:
    // 2.2 GB fetched per launch against 0.29 GB of input, profiles/r01_kernel_hbm_traffic.json)
    const int gy = Cout / BN;
    const long gx = (M + CV_BM - 1) / CV_BM;
    const long T = gx * gy, per_xcd = (T + 7) / 8;
    const long tl = (long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if ((long)(blockIdx.x >> 3) >= per_xcd || tl >= T) return;
    const long m0 = (tl / gy) * CV_BM;
    const int n0 = (int)(tl % gy) * BN;
    const int K = 9 * Cin;
    const int ksteps = K / CV_BK;

    // ---- A gather bookkeeping: this thread fetches pixels ra = tid/4 and ra+64, 32 contiguous bytes (2 chunks) at
    //      channel offset (tid%4)*16 of the 64-channel K slice; B: weight rows tid/4 (+64), the same 32 bytes ----
    const int part = (tid & 3) * 8 * CPT;
    const long HWd = (long)H * Wd;
    const long ma = m0 + (tid >> 2), mb = ma + 64;
    const bool ok0 = ma < M, ok1 = mb < M;
    const long mma = ok0 ? ma : 0, mmb = ok1 ? mb : 0;
    const long ba = mma / HWd, bb = mmb / HWd;
    const int rema = (int)(mma - ba * HWd), remb = (int)(mmb - bb * HWd);
    const int ay0 = rema / Wd, ax0 = rema - ay0 * Wd, ay1 = remb / Wd, ax1 = remb - ay1 * Wd;
    const long ab0 = ba * HWd, ab1 = bb * HWd;  // pixel index of (b, 0, 0)

    // staged registers (kept as named scalars: arrays captured by lambdas end up in scratch with hipcc)
    uint4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
#define CV_LOAD_TILES(KS)                                                                                              \
    {                                                                                                                  \
        const int k0_ = (KS) * CV_BK;                                                                                  \
        const int tap_ = k0_ / Cin, c0_ = k0_ - tap_ * Cin;                                                            \
        const int dy_ = tap_ / 3 - 1, dx_ = tap_ - (tap_ / 3) * 3 - 1;                                                 \
        {                                                                                                              \
            const int yy = ay0 + dy_, xx = ax0 + dx_;                                                                  \
            const bool ok = ok0 && yy >= 0 && yy < H && xx >= 0 && xx < Wd;                                            \
            const __hip_bfloat16 *p_ = X + ((ab0 + (ok ? (long)yy * Wd + xx : 0L)) * Cin + c0_ + part);                \
            ra0 = cv_load16_or_zero(p_, ok);                                                                           \
            if (CPT > 1) ra1 = cv_load16_or_zero(p_ + 8, ok);                                                                     \
        }                                                                                                              \
        {                                                                                                              \
            const int yy = ay1 + dy_, xx = ax1 + dx_;                                                                  \
            const bool ok = ok1 && yy >= 0 && yy < H && xx >= 0 && xx < Wd;                                            \
            const __hip_bfloat16 *p_ = X + ((ab1 + (ok ? (long)yy * Wd + xx : 0L)) * Cin + c0_ + part);                \
            ra2 = cv_load16_or_zero(p_, ok);                                                                           \
            if (CPT > 1) ra3 = cv_load16_or_zero(p_ + 8, ok);                                                                     \
        }                                                                                                              \
        {                                                                                                              \
            const __hip_bfloat16 *q_ = Wp + ((long)(n0 + (tid >> 2)) * K + k0_ + part);                                \
            rb0 = *reinterpret_cast<const uint4 *>(q_);                                                                \
            if (CPT > 1) rb1 = *reinterpret_cast<const uint4 *>(q_ + 8);                                                          \
            if (BN > 64) {                                                                                             \
                rb2 = *reinterpret_cast<const uint4 *>(q_ + 64L * K);                                                  \
                if (CPT > 1) rb3 = *reinterpret_cast<const uint4 *>(q_ + 64L * K + 8);                                            \
            }                                                                                                          \
        }                                                                                                              \
    }
#define CV_STORE_TILES(BUF)                                                                                            \
    {                                                                                                                  \
        const int sp_ = SWZ ? 8 * ((tid & 3) ^ ((tid >> 4) & 3)) : part;                                              \
        short *A_ = lds[BUF] + (tid >> 2) * CV_PITCH + sp_, *B_ = lds[BUF] + (CV_BM + (tid >> 2)) * CV_PITCH + sp_;    \
        *reinterpret_cast<uint4 *>(A_) = ra0;                                                                          \
        if (CPT > 1) *reinterpret_cast<uint4 *>(A_ + 8) = ra1;                                                                    \
        *reinterpret_cast<uint4 *>(A_ + 64 * CV_PITCH) = ra2;                                                          \
        if (CPT > 1) *reinterpret_cast<uint4 *>(A_ + 64 * CV_PITCH + 8) = ra3;                                                    \
        *reinterpret_cast<uint4 *>(B_) = rb0;                                                                          \
        if (CPT > 1) *reinterpret_cast<uint4 *>(B_ + 8) = rb1;                                                                    \
        if (BN > 64) {                                                                                                 \
            *reinterpret_cast<uint4 *>(B_ + 64 * CV_PITCH) = rb2;                                                      \
            if (CPT > 1) *reinterpret_cast<uint4 *>(B_ + 64 * CV_PITCH + 8) = rb3;                                                \
        }                                                                                                              \
    }
    ra1 = ra3 = rb1 = rb2 = rb3 = make_uint4(0u, 0u, 0u, 0u);

    f32x16 acc[2][NT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    CV_LOAD_TILES(0)
    CV_STORE_TILES(0)
    __syncthreads();

    const int frow = lane & 31, fk = (lane >> 5) * 8;  // fragment: row/col = lane%32, k offset = 8*(lane/32)
    const int fsw = (frow >> 2) & 3;                   // swizzle key of this lane's fragment row (tile bases are multiples of 16 rows)
    for (int ks = 0; ks < ksteps; ++ks) {
        const int cur = ks & 1;
        if (ks + 1 < ksteps) CV_LOAD_TILES(ks + 1)  // global -> VGPR, lands under the MFMAs below
        const short *A = lds[cur] + (wm * 64) * CV_PITCH;
        const short *Bs = lds[cur] + CV_BM * CV_PITCH + (wn * WN) * CV_PITCH;
#pragma unroll
        for (int kk = 0; kk < CV_BK; kk += 16) {
            bf16x8 af[2], bfr[NT];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = *reinterpret_cast<const bf16x8 *>(A + (i * 32 + frow) * CV_PITCH + (SWZ ? 8 * ((((kk + fk) >> 3)) ^ fsw) : kk + fk));
#pragma unroll
            for (int j = 0; j < NT; ++j) bfr[j] = *reinterpret_cast<const bf16x8 *>(Bs + (j * 32 + frow) * CV_PITCH + (SWZ ? 8 * ((((kk + fk) >> 3)) ^ fsw) : kk + fk));
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        }
        if (ks + 1 < ksteps) CV_STORE_TILES(cur ^ 1)
        __syncthreads();
    }

    // ---- epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) ----
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = n0 + wn * WN + j * 32 + (lane & 31);
        const float bv = bias ? bias[n] : 0.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                float v = acc[i][j][r] + bv;
                if (RELU) v = v > 0.0f ? v : 0.0f;
                if (m < M) Y[m * Cout + n] = __float2bfloat16(v);
            }
        }
    }
}

#undef CV_LOAD_TILES
#undef CV_STORE_TILES

// W[n][c][ky][kx] (fp32 or bf16 source, given as fp32 here) -> Wp[n][(ky*3+kx)*Cin + c] bf16          (forward)
//                                                           -> Wp'[c][((2-ky)*3+(2-kx))*Cout + n]     (data gradient)
__global__ __launch_bounds__(256) void pack_conv3x3_weights_kernel(const float *__restrict__ W, int Cout, int Cin, int transpose_flip,
                                                                   __hip_bfloat16 *__restrict__ Wp) {
    const long total = (long)Cout * Cin * 9;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int kx = (int)(i % 3), ky = (int)((i / 3) % 3);
    const int c = (int)((i / 9) % Cin), n = (int)(i / (9L * Cin));
    const float v = W[i];
    long o;
    if (!transpose_flip) o = (long)n * 9 * Cin + (ky * 3 + kx) * Cin + c;
    else o = (long)c * 9 * Cout + ((2 - ky) * 3 + (2 - kx)) * Cout + n;
    Wp[o] = __float2bfloat16(v);
}

extern "C" int xq_conv3x3_pack_weights(const float *W, int Cout, int Cin, int for_data_grad, void *Wp, xq_stream_t stream) {
    if (!W || !Wp || Cout < 1 || Cin < 1) return xq_set_error(XQ_EINVAL, "%s: bad arguments", "xq_conv3x3_pack_weights");
    const long total = (long)Cout * Cin * 9;
    hipLaunchKernelGGL(pack_conv3x3_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W, Cout, Cin,
                       for_data_grad, (__hip_bfloat16 *)Wp);
    return xq_check_launch("pack_conv3x3_weights_kernel");
}

// Every registered 3x3 weight of an arena in ONE launch behind the optimizer step (round 6: the CNN tokenizer repacked 69 weights x 2 layouts
// with 138 launches of 8 us per train step).  table: DEVICE int64 [n][6] = {fp32 source W, forward pack or 0, data-gradient pack or 0 (device
// addresses), Cout, Cin, first block}; a weight owns ceil(Cout * Cin * 9 / 256) consecutive blocks.
__global__ __launch_bounds__(256) void pack_conv3x3_weights_batched_kernel(const long *__restrict__ table, int n) {
    const long t = blockIdx.x;
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[6 * mid + 5] <= t) lo = mid; else hi = mid - 1;
    }
    const long *e = table + 6 * lo;
    const float *W = reinterpret_cast<const float *>(e[0]);
    __hip_bfloat16 *Wf = reinterpret_cast<__hip_bfloat16 *>(e[1]), *Wd = reinterpret_cast<__hip_bfloat16 *>(e[2]);
    const int Cout = (int)e[3], Cin = (int)e[4];
    const long i = (t - e[5]) * 256 + threadIdx.x;
    if (i >= (long)Cout * Cin * 9) return;
    const int kx = (int)(i % 3), ky = (int)((i / 3) % 3);
    const int c = (int)((i / 9) % Cin), nn = (int)(i / (9L * Cin));
    const __hip_bfloat16 v = __float2bfloat16(W[i]);
    if (Wf) Wf[(long)nn * 9 * Cin + (ky * 3 + kx) * Cin + c] = v;
    if (Wd) Wd[(long)c * 9 * Cout + ((2 - ky) * 3 + (2 - kx)) * Cout + nn] = v;
}

extern "C" int xq_conv3x3_pack_weights_batched(const int64_t *table, int n_weights, int64_t blocks, xq_stream_t stream) {
    const char *fn = "xq_conv3x3_pack_weights_batched";
    if (n_weights < 0 || blocks < 0) return xq_set_error(XQ_EINVAL, "%s: negative count", fn);
    if (n_weights == 0 || blocks == 0) return XQ_OK;
    if (!table) return xq_set_error(XQ_EINVAL, "%s: null table", fn);
    if (blocks > 0x7fffffffL) return xq_set_error(XQ_EINVAL, "%s: %ld blocks exceed the grid", fn, (long)blocks);
    static_assert(sizeof(long) == sizeof(int64_t), "table entries are read as long");
    hipLaunchKernelGGL(pack_conv3x3_weights_batched_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const long *)table, n_weights);
    return xq_check_launch("pack_conv3x3_weights_batched_kernel");
}

// ---------------------------------------------------------------------------------------------------------------------
// 64 -> 64 channels (conv1_2 of the LPIPS VGG16 trunk at 256^2 x 128 images, forward and data gradient: 4.3 GB of activations per
// call against 0.6 TFLOP — the layer is HBM-bound on paper, 0.45 ms at 5 TB/s, and took 1.47 ms on the 128-pixel kernel above, which
// re-gathers every input pixel nine times through registers with a barrier per 32-deep K step).  Round 4:
//   * the whole weight matrix [9 taps][64 cout][64 cin] bf16 = 72 KiB stays in LDS for the life of a persistent workgroup (one per CU);
//   * an output tile is 16 rows x 32 pixels; its 18 x 34 input halo arrives ONCE by LDS-DMA (global_load_lds_dwordx4, out-of-image
//     pixels from a zero page) as 128-byte pixel slots, and all nine taps read their A fragments from it at shifted slot indices
//     (measured and dropped: 8 x 32 tiles with TWO halo buffers, the next halo landing under the compute — 0.96 ms against 0.88 ms:
//     12 fragment reads per 8 MFMAs instead of 16 per 16, profiles/r04_conv_c64.txt);
//   * 8 waves, wave w owns output rows 2w, 2w + 1: per tap 8 A + 8 B fragment reads feed 16 MFMAs (no barrier inside a tile).
// LDS rows of 128 B (weights: row = tap * 64 + cout; halo: row = slot = r * 34 + c): 16-byte chunk c sits at position
// c ^ ((row >> 1) & 7), applied to the DMA's source address and on the reads — conflict-free ds_read_b128 for 32 consecutive rows at
// any even or odd base (the shifted taps).
// ---------------------------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void cv_lds_void;
typedef __attribute__((address_space(1))) void cv_gbl_void;
__device__ __attribute__((aligned(64))) char cv_zero_page[64];

static constexpr int C64_TH = 16, C64_TW = 32, C64_HW = C64_TW + 2, C64_SLOTS = (C64_TH + 2) * (C64_TW + 2);     // 612 halo slots
static constexpr int C64_W_BYTES = 9 * 64 * 128, C64_H_INSTR = (C64_SLOTS + 7) / 8, C64_LDS = C64_W_BYTES + C64_H_INSTR * 1024;   // 72 + 77 KiB

template <bool RELU>
__global__ __launch_bounds__(512, 2) void conv3x3_c64_kernel(const char *__restrict__ X, const char *__restrict__ Wp, const float *__restrict__ bias,
                                                          int H, int Wd, char *__restrict__ Y, int tiles_y, int tiles_x, long ntiles,
                                                          const char *__restrict__ Mk) {
    extern __shared__ __attribute__((aligned(16))) char c64_smem[];
    char *const Wl = c64_smem, *const Hl = c64_smem + C64_W_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, hh = lane >> 5;
    const int drow = lane >> 3, dcp = lane & 7;

    // weights -> LDS, once: wave-instruction i moves LDS rows 8 i .. 8 i + 7
    for (int i = wave; i < 72; i += 8) {
        const int r = 8 * i + drow;
        const int c = dcp ^ ((r >> 1) & 7);
        const char *src = Wp + ((long)(r & 63) * 576 + (r >> 6) * 64 + 8 * c) * 2;
        __builtin_amdgcn_global_load_lds((cv_gbl_void *)src, (cv_lds_void *)(Wl + i * 1024), 16, 0, 0);
    }
    // B fragment offsets (row = 32 cg + li inside a tap: (row >> 1) & 7 = (li >> 1) & 7)
    unsigned bo[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) bo[k] = (unsigned)(li * 128 + (((2 * k + hh) ^ ((li >> 1) & 7)) << 4));
    // this lane's 32 bias values: couts 32 cg + 8 q + 4 hh + 0..3
    float4 bv[2][4];
#pragma unroll
    for (int cg = 0; cg < 2; ++cg)
#pragma unroll
        for (int q = 0; q < 4; ++q)
            bv[cg][q] = bias ? *reinterpret_cast<const float4 *>(bias + 32 * cg + 8 * q + 4 * hh) : make_float4(0.f, 0.f, 0.f, 0.f);

    const long per_img = (long)tiles_y * tiles_x;
    for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int b = (int)(tile / per_img);
        const int tr = (int)(tile - (long)b * per_img);
        const int y0 = (tr / tiles_x) * C64_TH, x0 = (tr % tiles_x) * C64_TW;
        // every wave is done reading the previous tile's halo (its fragment reads fed MFMAs that have issued).  A raw barrier: __syncthreads()
        // would also wait for the previous tile's output stores (vmcnt(0)) before the halo DMA may even be issued
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        for (int i = wave; i < C64_H_INSTR; i += 8) {
            const int s = 8 * i + drow;
            const int hr = s / C64_HW, hc = s - hr * C64_HW;
            const int yy = y0 - 1 + hr, xx = x0 - 1 + hc;
            const bool ok = s < C64_SLOTS && yy >= 0 && yy < H && xx >= 0 && xx < Wd;
            const int c = dcp ^ ((s >> 1) & 7);
            const char *src = ok ? X + (((long)b * H + yy) * Wd + xx) * 128 + 16 * c : cv_zero_page;
            __builtin_amdgcn_global_load_lds((cv_gbl_void *)src, (cv_lds_void *)(Hl + i * 1024), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int ky = tap / 3, kx = tap - 3 * ky;
            bf16x8 bf[2][4];
#pragma unroll
            for (int cg = 0; cg < 2; ++cg)
#pragma unroll
                for (int k = 0; k < 4; ++k) bf[cg][k] = *reinterpret_cast<const bf16x8 *>(Wl + tap * 8192 + cg * 4096 + bo[k]);
#pragma unroll
            for (int pg = 0; pg < 2; ++pg) {
                const int s = (2 * wave + pg + ky) * C64_HW + li + kx;
                const int sw = (s >> 1) & 7;
                bf16x8 af[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) af[k] = *reinterpret_cast<const bf16x8 *>(Hl + s * 128 + (((2 * k + hh) ^ sw) << 4));
#pragma unroll
                for (int cg = 0; cg < 2; ++cg)
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[pg][cg] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bf[cg][k], af[k], acc[pg][cg], 0, 0, 0);
            }
        }
        // epilogue: lane (li, hh) holds pixel li of its two rows, couts 32 cg + 8 q + 4 hh + 0..3 in register quad q.  The two half-waves
        // exchange quads (v_permlane32_swap: cdna_hip_programming.md T21) so that every lane stores 16 contiguous bytes: lanes 0-31 the
        // couts 8 q .. 8 q + 7 of the even quad, lanes 32-63 those of the odd quad — 8 instead of 16 store instructions per wave and tile
#pragma unroll
        for (int pg = 0; pg < 2; ++pg) {
            const int yy = y0 + 2 * wave + pg, xx = x0 + li;
            const bool ok = yy < H && xx < Wd;
            char *yp = Y + (((long)b * H + (ok ? yy : 0)) * Wd + (ok ? xx : 0)) * 128 + 16 * hh;
#pragma unroll
            for (int cg = 0; cg < 2; ++cg) {
                uint2 pk[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float e0 = acc[pg][cg][4 * q + 0] + bv[cg][q].x, e1 = acc[pg][cg][4 * q + 1] + bv[cg][q].y;
                    float e2 = acc[pg][cg][4 * q + 2] + bv[cg][q].z, e3 = acc[pg][cg][4 * q + 3] + bv[cg][q].w;
                    if (RELU) { e0 = fmaxf(e0, 0.f); e1 = fmaxf(e1, 0.f); e2 = fmaxf(e2, 0.f); e3 = fmaxf(e3, 0.f); }
                    typedef __bf16 cv_bf2 __attribute__((ext_vector_type(2)));
                    typedef float cv_f2 __attribute__((ext_vector_type(2)));
                    pk[q].x = __builtin_bit_cast(unsigned, __builtin_convertvector(cv_f2{e0, e1}, cv_bf2));
                    pk[q].y = __builtin_bit_cast(unsigned, __builtin_convertvector(cv_f2{e2, e3}, cv_bf2));
                }
#pragma unroll
                for (int q = 0; q < 4; q += 2) {
                    auto rx = __builtin_amdgcn_permlane32_swap(pk[q].x, pk[q + 1].x, false, false);
                    auto ry = __builtin_amdgcn_permlane32_swap(pk[q].y, pk[q + 1].y, false, false);
                    uint4 o = make_uint4(rx[0], ry[0], rx[1], ry[1]);
                    if (Mk != nullptr && ok) {      // out_mask: same layout as Y
                        const uint4 mk = *reinterpret_cast<const uint4 *>(Mk + (yp - Y) + (32 * cg + 8 * q) * 2);
                        o.x = xq::keep_where_positive(o.x, mk.x); o.y = xq::keep_where_positive(o.y, mk.y);
                        o.z = xq::keep_where_positive(o.z, mk.z); o.w = xq::keep_where_positive(o.w, mk.w);
                    }
                    if (ok) *reinterpret_cast<uint4 *>(yp + (32 * cg + 8 * q) * 2) = o;
                }
            }
        }
    }
}

static bool c64_disabled() {
    static const bool off = [] { const char *e = getenv("XQ_CONV_C64"); return e && e[0] == '0'; }();
    return off;
}
extern "C" int xq_conv3x3_nhwc_bf16_takes_out_mask(int Cin, int Cout) { return Cin == 64 && Cout == 64 && !c64_disabled(); }

extern "C" int xq_conv3x3_nhwc_bf16(const void *X, const void *Wp, const float *bias, int B, int H, int W, int Cin, int Cout, int relu,
                                    const void *out_mask, void *Y, xq_stream_t stream) {
    const char *fn = "xq_conv3x3_nhwc_bf16";
    if (B == 0) return XQ_OK;
    if (!X || !Wp || !Y) return xq_set_error(XQ_EINVAL, "%s: null pointer", fn);
    if (Cin % 64 != 0 || Cout % 64 != 0)
        return xq_set_error(XQ_EINVAL, "%s: needs Cin %% 64 == 0 and Cout %% 64 == 0 (got %ld, %ld)", fn, Cin, Cout);
    if (out_mask && !xq_conv3x3_nhwc_bf16_takes_out_mask(Cin, Cout))
        return xq_set_error(XQ_EINVAL, "%s: out_mask is folded into the store of the 64 -> 64 channel kernel only (got %ld -> %ld)", fn, Cin, Cout);
    const long M = (long)B * H * W;
    hipStream_t s = (hipStream_t)stream;
    const long gx = (M + CV_BM - 1) / CV_BM;
    const __hip_bfloat16 *x = (const __hip_bfloat16 *)X, *w = (const __hip_bfloat16 *)Wp;
    __hip_bfloat16 *y = (__hip_bfloat16 *)Y;
    const int pslot = prof_begin(XQ_PROF_CONV3X3, 2.0 * (double)M * 9.0 * Cin * Cout, s);
    if (Cin == 64 && Cout == 64 && !c64_disabled()) {      // weights resident in LDS, one halo load per 16 x 32 output tile
        static unsigned long long attr_devs = 0;      // per device (a process driving several GPUs)
        if (first_call_on_this_device(&attr_devs)) {
            if (hipFuncSetAttribute(reinterpret_cast<const void *>(conv3x3_c64_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, C64_LDS) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void *>(conv3x3_c64_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, C64_LDS) != hipSuccess)
                return xq_set_error(XQ_ELAUNCH, "%s: hipFuncSetAttribute failed", fn);
        }
        const int ty = (H + C64_TH - 1) / C64_TH, tx = (W + C64_TW - 1) / C64_TW;
        const long ntiles = (long)B * ty * tx;
        const long grid = ntiles < num_cus() ? ntiles : num_cus();
        if (relu) hipLaunchKernelGGL((conv3x3_c64_kernel<true>), dim3((unsigned)grid), dim3(512), C64_LDS, s, (const char *)X, (const char *)Wp, bias, H, W, (char *)Y, ty, tx, ntiles, (const char *)out_mask);
        else hipLaunchKernelGGL((conv3x3_c64_kernel<false>), dim3((unsigned)grid), dim3(512), C64_LDS, s, (const char *)X, (const char *)Wp, bias, H, W, (char *)Y, ty, tx, ntiles, (const char *)out_mask);
    } else if (Cout % 128 == 0) {
        if (relu) hipLaunchKernelGGL((conv3x3_kernel<128, 64, true>), dim3((unsigned)(((gx * (Cout / 128) + 7) / 8) * 8)), dim3(256), 0, s, x, w, bias, M, H, W, Cin, Cout, y);
        else hipLaunchKernelGGL((conv3x3_kernel<128, 64, false>), dim3((unsigned)(((gx * (Cout / 128) + 7) / 8) * 8)), dim3(256), 0, s, x, w, bias, M, H, W, Cin, Cout, y);
    } else {
        if (relu) hipLaunchKernelGGL((conv3x3_kernel<64, 32, true>), dim3((unsigned)(((gx * (Cout / 64) + 7) / 8) * 8)), dim3(256), 0, s, x, w, bias, M, H, W, Cin, Cout, y);
        else hipLaunchKernelGGL((conv3x3_kernel<64, 32, false>), dim3((unsigned)(((gx * (Cout / 64) + 7) / 8) * 8)), dim3(256), 0, s, x, w, bias, M, H, W, Cin, Cout, y);
    }
    prof_end(pslot, s);
    return xq_check_launch(fn);
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient of the 3x3 convolution: dWp[n][tap*Cin + c] += sum_m dY[m][n] * X[m shifted by tap][c]  (fp32).
// GEMM view per tap: rows n (Cout), columns c (Cin), reduction over the pixels m.  Both operands are stored pixel-major
// ([m][channels], NHWC), i.e. with the REDUCTION index as the slow axis; the MFMA wants 8 consecutive reduction indices
// per lane, so both fragments come out of the row-major LDS tiles through the transpose read ds_read_b64_tr_b16 (lane =
// channel, 4 consecutive pixels per read; the same register <-> pixel order on both operands).
// Block = one tap x 128 n x 128 c over a contiguous range of 64-pixel tiles; 2 x 2 waves of 64 x 64; the partial tile is
// added to dWp with fp32 atomics (dWp zero-initialised by the caller).  The shifted X rows use the forward's halo logic.
// Staging goes global -> registers (issued before the tile's MFMAs) -> LDS; NBUF = 2: the store lands in the other buffer, one barrier per K
// tile; NBUF = 1 (the product form, see the launch plan below): one 37 KiB buffer, a second barrier per K tile, four workgroups per CU.
// ---------------------------------------------------------------------------------------------------------------------
static inline bool is_pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }
typedef short wg_s4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ wg_s4 wg_tr4(const short *p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((wg_s4 __attribute__((address_space(3))) *)p);
}
// WG_PITCH (template parameter): LDS row pitch in bf16 elements.  A 16-lane group of ds_read_b64_tr_b16 reads 4 rows x 32 bytes: the rows'
// segments fall into disjoint banks when the pitch is 32 bytes mod 256 (144 elements = 288 bytes); round 2's 136 (272 bytes = 16 mod 256)
// made neighbouring rows overlap by half — 2-way conflicts on every fragment read.  XQ_WGRAD_PITCH=136 keeps the old pitch for A/B timing.
#define WG_TFRAG(TRP, ROW0, COL0) __builtin_shufflevector(wg_tr4((TRP) + (ROW0) * WG_PITCH + (COL0)), wg_tr4((TRP) + ((ROW0) + 8) * WG_PITCH + (COL0)), 0, 1, 2, 3, 4, 5, 6, 7)

// Geometry: dY pixel (b, y, x) of an H x Wd map pairs with X pixel ((y * stride + ky - pad) >> up, (x * stride + kx - pad) >> up) of an
// Hi x Wi map (stride 2 / pad 0: Downsample; up = 1: the conv ran on the nearest-2x upsampled X; default stride 1 / pad 1 / Hi = H).
// MODE 0: any map size — pixel -> (b, y, x) by two 32-bit divisions per staged row, 8 per thread and K tile: ~280 VALU instructions in the
//         load block against the tile's 16 MFMAs per wave (the kernel is VALU-bound in this form).
// MODE 1: H and Wd powers of two: shifts and masks (~150 VALU instructions).
// MODE 2: additionally Wd >= 16, M % 64 == 0 and both tensors below 4 GiB (every map of the 256 x 256 CNN tokenizer).  The 16 rows a
//         thread group stages per step (srow = 0..15 at one `I`) lie in ONE image row, so batch index, y, the row's validity and the row base
//         are wave-uniform (scalar ALU); per lane only x's validity and a 32-bit byte offset remain, and out-of-image taps become buffer
//         loads past the descriptor's range (they return zero): ~25 VALU instructions per K tile.
typedef unsigned wg_u32x4 __attribute__((ext_vector_type(4)));
template <int MODE, int WG_PITCH, int NBUF = 2, int MINW = 1>
__global__ __launch_bounds__(256, MINW) void conv3x3_wgrad_kernel(const __hip_bfloat16 *__restrict__ X, const __hip_bfloat16 *__restrict__ dY, long M,
                                                            int H, int Wd, int Cin, int Cout, int tiles_per_split, int nsplit,
                                                            float *__restrict__ dWp, int Hi, int Wi, int stride, int pad, int up,
                                                            unsigned x_bytes, unsigned y_bytes) {
    constexpr bool POW2 = MODE >= 1;
    __shared__ __attribute__((aligned(16))) short lds[NBUF][2 * 64 * WG_PITCH];   // [buffer][dY tile | X tile], 64 pixels each
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wc = wave & 1;
    const int gn = Cout / 128, gc = Cin / 128;
    // XCD-aware order: XCD k walks a contiguous range of work items; within a pixel split the 9 taps x channel tiles run
    // back to back, so the dY / X rows of that pixel range stay in one L2
    const long T = (long)nsplit * 9 * gn * gc, per_xcd = (T + 7) / 8;
    const long tl = (long)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (tl >= T) return;
    int rest = (int)(tl % (9 * gn * gc));
    const int split = (int)(tl / (9 * gn * gc));
    const int tap = rest / (gn * gc);
    rest -= tap * gn * gc;
    const int n0 = (rest / gc) * 128, c0 = (rest % gc) * 128;
    const int dy = tap / 3 - pad, dx = tap - (tap / 3) * 3 - pad;
    const int Hl = up ? 2 * Hi : Hi, Wl = up ? 2 * Wi : Wi;
    const long HWin = (long)Hi * Wi;
    const long ntiles = (M + 63) / 64;
    const long t_begin = (long)split * tiles_per_split;
    long t_end = t_begin + tiles_per_split;
    if (t_end > ntiles) t_end = ntiles;
    if (t_begin >= t_end) return;

    // staging: thread -> rows (tid/16) + 16*i (i = 0..3) of the 64-pixel tile, 16-byte chunk tid%16 of the 128 channels
    const int srow = tid >> 4, spart = (tid & 15) * 8;
    const int HWi = H * Wd;
    const int lw = 31 - __builtin_clz((unsigned)Wd), lhw = 31 - __builtin_clz((unsigned)HWi);     // (POW2 only)
    uint4 ry0, ry1, ry2, ry3, rx0, rx1, rx2, rx3;
    const uint4 zero4 = make_uint4(0u, 0u, 0u, 0u);
#define WG_LOAD_ROW(RY, RX, I, TILE)                                                                                   \
    {                                                                                                                  \
        const long m_ = (TILE) * 64 + srow + 16 * (I);                                                                 \
        const bool okm = m_ < M;                                                                                       \
        const long mm = okm ? m_ : 0;                                                                                  \
        RY = *reinterpret_cast<const uint4 *>(dY + mm * Cout + n0 + spart);                                            \
        if (!okm) RY = zero4;                                                                                          \
        int b_, y_, x_;                           /* M < 2^31 (checked by the launcher): 32-bit arithmetic */         \
        if (POW2) {                                                                                                    \
            b_ = (int)mm >> lhw; y_ = ((int)mm >> lw) & (H - 1); x_ = (int)mm & (Wd - 1);                              \
        } else {                                                                                                       \
            b_ = (int)mm / HWi;                                                                                        \
            const int rem_ = (int)mm - b_ * HWi;                                                                       \
            y_ = rem_ / Wd; x_ = rem_ - y_ * Wd;                                                                       \
        }                                                                                                              \
        const int yy = y_ * stride + dy, xx = x_ * stride + dx;                                                        \
        const bool ok = okm && yy >= 0 && yy < Hl && xx >= 0 && xx < Wl;                                               \
        RX = *reinterpret_cast<const uint4 *>(X + (((long)b_ * HWin + (ok ? (yy >> up) * Wi + (xx >> up) : 0)) * Cin + c0 + spart)); \
        if (!ok) RX = zero4;                                                                                           \
    }
    // MODE 2 (see above): descriptors over the whole tensors, per-lane constants of the offsets
    const auto rs_x = __builtin_amdgcn_make_buffer_rsrc((void *)X, 0, MODE == 2 ? x_bytes : 0u, 0x00020000);
    const auto rs_y = __builtin_amdgcn_make_buffer_rsrc((void *)dY, 0, MODE == 2 ? y_bytes : 0u, 0x00020000);
    const unsigned lane_y = (unsigned)(srow * Cout + spart) * 2u, lane_c = (unsigned)spart * 2u, cin2 = (unsigned)Cin * 2u;
    const int lane_xs = srow * stride;
#define WG_LOAD_ROW_FAST(RY, RX, I, TILE)                                                                              \
    {                                                                                                                  \
        const int m0_ = (int)(TILE) * 64 + 16 * (I);                  /* wave-uniform: first pixel of this 16-row group */ \
        const int b_ = m0_ >> lhw, y_ = (m0_ >> lw) & (H - 1), x0_ = m0_ & (Wd - 1);                                   \
        const int yy = y_ * stride + dy;                                                                               \
        const bool rowok = (unsigned)yy < (unsigned)Hl;                                                                \
        const unsigned sbase = ((unsigned)((b_ * Hi + (yy >> up)) * Wi) * (unsigned)Cin + (unsigned)c0) * 2u;          \
        const int xx = x0_ * stride + dx + lane_xs;                                                                    \
        unsigned vo = sbase + (unsigned)(xx >> up) * cin2 + lane_c;                                                    \
        vo = (rowok && (unsigned)xx < (unsigned)Wl) ? vo : 0xffffffffu;                                                \
        RX = __builtin_bit_cast(uint4, (wg_u32x4)__builtin_amdgcn_raw_buffer_load_b128(rs_x, vo, 0, 0));               \
        RY = __builtin_bit_cast(uint4, (wg_u32x4)__builtin_amdgcn_raw_buffer_load_b128(rs_y, ((unsigned)m0_ * (unsigned)Cout + (unsigned)n0) * 2u + lane_y, 0, 0)); \
    }
#define WG_LOAD(TILE)                                                                                                  \
    {                                                                                                                  \
        if (MODE == 2) {                                                                                               \
            WG_LOAD_ROW_FAST(ry0, rx0, 0, TILE) WG_LOAD_ROW_FAST(ry1, rx1, 1, TILE)                                    \
            WG_LOAD_ROW_FAST(ry2, rx2, 2, TILE) WG_LOAD_ROW_FAST(ry3, rx3, 3, TILE)                                    \
        } else {                                                                                                       \
            WG_LOAD_ROW(ry0, rx0, 0, TILE) WG_LOAD_ROW(ry1, rx1, 1, TILE) WG_LOAD_ROW(ry2, rx2, 2, TILE) WG_LOAD_ROW(ry3, rx3, 3, TILE) \
        }                                                                                                              \
    }
#define WG_STORE(BUF)                                                                                                  \
    {                                                                                                                  \
        short *Yt = lds[BUF] + srow * WG_PITCH + spart, *Xt = Yt + 64 * WG_PITCH;                                      \
        *reinterpret_cast<uint4 *>(Yt) = ry0;                                                                          \
        *reinterpret_cast<uint4 *>(Yt + 16 * WG_PITCH) = ry1;                                                          \
        *reinterpret_cast<uint4 *>(Yt + 32 * WG_PITCH) = ry2;                                                          \
        *reinterpret_cast<uint4 *>(Yt + 48 * WG_PITCH) = ry3;                                                          \
        *reinterpret_cast<uint4 *>(Xt) = rx0;                                                                          \
        *reinterpret_cast<uint4 *>(Xt + 16 * WG_PITCH) = rx1;                                                          \
        *reinterpret_cast<uint4 *>(Xt + 32 * WG_PITCH) = rx2;                                                          \
        *reinterpret_cast<uint4 *>(Xt + 48 * WG_PITCH) = rx3;                                                          \
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int hh = lane >> 5;
    const int troff = (4 * hh + ((lane & 15) >> 2)) * WG_PITCH + 16 * ((lane >> 4) & 1) + 4 * (lane & 3);
    WG_LOAD(t_begin)
    WG_STORE(0)
    __syncthreads();
    for (long t = t_begin; t < t_end; ++t) {
        const int cur = NBUF == 2 ? (int)((t - t_begin) & 1) : 0;
        if (t + 1 < t_end) WG_LOAD(t + 1)
        const short *Yt = lds[cur] + troff + wn * 64, *Xt = lds[cur] + 64 * WG_PITCH + troff + wc * 64;
#pragma unroll
        for (int kk = 0; kk < 64; kk += 16) {
            const bf16x8 a0 = WG_TFRAG(Yt, kk, 0), a1 = WG_TFRAG(Yt, kk, 32);
            const bf16x8 b0 = WG_TFRAG(Xt, kk, 0), b1 = WG_TFRAG(Xt, kk, 32);
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
        }
        if (NBUF == 1) __syncthreads();      // one buffer: every wave is done reading before the next tile overwrites it
        if (t + 1 < t_end) WG_STORE(NBUF == 2 ? (cur ^ 1) : 0)
        __syncthreads();
    }
#undef WG_LOAD_ROW
#undef WG_LOAD_ROW_FAST
#undef WG_LOAD
#undef WG_STORE

    // D[i = n][j = c]: column c = lane & 31, row n = (r & 3) + 8 * (r >> 2) + 4 * hh
    const long K9 = 9L * Cin;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = c0 + wc * 64 + j * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wn * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                atomicAdd(dWp + (long)n * K9 + (long)tap * Cin + c, acc[i][j][r]);
            }
        }
}

// Launch plan.  The product form for the maps of the CNN tokenizer (MODE 2, one LDS buffer, 128 VGPRs: four workgroups = 16 waves per CU —
// round 5: with two buffers the 74 KiB of LDS held the kernel at two workgroups per CU, and its one barrier per 16 MFMAs had nothing to hide
// behind; one buffer costs a second barrier per K tile and wins 15-35 %, profiles/r05_conv_wgrad_ab.txt) runs ~8 workgroups per CU in total, but
// never fewer than 32 K tiles per workgroup (the 64 atomics per lane of the epilogue); the general forms keep two buffers and ~4 per CU.
// Environment switches for A/B timing: XQ_WGRAD_TWO_BUFFERS=1 (MODE 2 with two buffers), XQ_WGRAD_PITCH=136 (round-2 LDS pitch, two-buffer forms),
// XQ_WGRAD_SLOW_INDEX=1 (MODE 1 instead of 2), XQ_WGRAD_BLOCKS_PER_CU=n.
static int conv3x3_wgrad_launch(const char *fn, const void *X, const void *dY, int B, int Hi, int Wi, int Ho, int Wo, int Cin, int Cout, int stride,
                                int pad, int up, float *dWp, hipStream_t s) {
    static const bool old_pitch = getenv("XQ_WGRAD_PITCH") && atoi(getenv("XQ_WGRAD_PITCH")) == 136;
    static const bool two_buf = getenv("XQ_WGRAD_TWO_BUFFERS") != nullptr || old_pitch;
    static const bool slow_index = getenv("XQ_WGRAD_SLOW_INDEX") != nullptr;
    static const long per_cu_env = getenv("XQ_WGRAD_BLOCKS_PER_CU") ? atol(getenv("XQ_WGRAD_BLOCKS_PER_CU")) : 0L;
    const long M = (long)B * Ho * Wo;
    const long xb = (long)B * Hi * Wi * Cin * 2, yb = M * Cout * 2;
    const bool p2 = is_pow2(Ho) && is_pow2(Wo);
    const int mode = (p2 && Wo >= 16 && M % 64 == 0 && xb < 0xffffff00L && yb < 0xffffff00L && !slow_index) ? 2 : (p2 ? 1 : 0);
    const bool one_buf = mode == 2 && !two_buf;
    const long ntiles = (M + 63) / 64;
    const long base = 9L * (Cout / 128) * (Cin / 128);
    const long per_cu = per_cu_env > 0 ? per_cu_env : (one_buf ? 8L : 4L);
    long nsplit = (per_cu * num_cus() + base - 1) / base;
    if (one_buf && nsplit > ntiles / 32) nsplit = ntiles / 32;
    if (nsplit < 1) nsplit = 1;
    if (nsplit > ntiles) nsplit = ntiles;
    const long tps = (ntiles + nsplit - 1) / nsplit;
    nsplit = (ntiles + tps - 1) / tps;
    const long T = nsplit * base;
    const dim3 grid((unsigned)(((T + 7) / 8) * 8));
#define WG_GO(...)                                                                                                                        \
    hipLaunchKernelGGL((conv3x3_wgrad_kernel<__VA_ARGS__>), grid, dim3(256), 0, s, (const __hip_bfloat16 *)X, (const __hip_bfloat16 *)dY, M, Ho, Wo, \
                       Cin, Cout, (int)tps, (int)nsplit, dWp, Hi, Wi, stride, pad, up, (unsigned)xb, (unsigned)yb)
    if (one_buf) WG_GO(2, 144, 1, 4);
    else if (mode == 2) { if (old_pitch) WG_GO(2, 136); else WG_GO(2, 144); }
    else if (mode == 1) { if (old_pitch) WG_GO(1, 136); else WG_GO(1, 144); }
    else { if (old_pitch) WG_GO(0, 136); else WG_GO(0, 144); }
#undef WG_GO
    return xq_check_launch(fn);
}

extern "C" int xq_conv3x3_wgrad_nhwc_bf16(const void *X, const void *dY, int B, int H, int W, int Cin, int Cout, float *dWp,
                                          xq_stream_t stream) {
    const char *fn = "xq_conv3x3_wgrad_nhwc_bf16";
    if (B == 0) return XQ_OK;
    if (!X || !dY || !dWp) return xq_set_error(XQ_EINVAL, "%s: null pointer", fn);
    if (Cin % 128 != 0 || Cout % 128 != 0)
        return xq_set_error(XQ_EINVAL, "%s: needs Cin %% 128 == 0 and Cout %% 128 == 0 (got %ld, %ld)", fn, Cin, Cout);
    if ((long)B * H * W >= (1L << 31)) return xq_set_error(XQ_EINVAL, "%s: B*H*W must be below 2^31", fn);
    return conv3x3_wgrad_launch(fn, X, dY, B, H, W, H, W, Cin, Cout, 1, 1, 0, dWp, (hipStream_t)stream);
}

extern "C" int xq_conv3x3_wgrad_nhwc_bf16_ex(const void *X, const void *dY, int B, int Hi, int Wi, int Ho, int Wo, int Cin, int Cout, int stride,
                                             int pad, int upsample2x, float *dWp, xq_stream_t stream) {
    const char *fn = "xq_conv3x3_wgrad_nhwc_bf16_ex";
    if (B == 0) return XQ_OK;
    if (!X || !dY || !dWp) return xq_set_error(XQ_EINVAL, "%s: null pointer", fn);
    if (Cin % 128 != 0 || Cout % 128 != 0)
        return xq_set_error(XQ_EINVAL, "%s: needs Cin %% 128 == 0 and Cout %% 128 == 0 (got %ld, %ld)", fn, Cin, Cout);
    if ((stride != 1 && stride != 2) || pad < 0 || pad > 1) return xq_set_error(XQ_EINVAL, "%s: stride 1 / 2, pad 0 / 1", fn);
    if ((long)B * Ho * Wo >= (1L << 31) || (long)B * Hi * Wi >= (1L << 31)) return xq_set_error(XQ_EINVAL, "%s: pixel counts must be below 2^31", fn);
    return conv3x3_wgrad_launch(fn, X, dY, B, Hi, Wi, Ho, Wo, Cin, Cout, stride, pad, upsample2x ? 1 : 0, dWp, (hipStream_t)stream);
}

// out[b][y][x][c] = sum of the 2 x 2 block of in at (2y, 2x): the backward of nearest-2x upsampling (Upsample, xqgan_model.py:682-686)
__global__ __launch_bounds__(256) void sumpool2x2_kernel(const __hip_bfloat16 *__restrict__ in, int B, int Ho, int Wo, int C,
                                                         __hip_bfloat16 *__restrict__ out) {
    const int cv = C / 8;
    const long total = (long)B * Ho * Wo * cv;
    const long Wi = 2L * Wo;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % cv);
        long t = i / cv;
        const int xo = (int)(t % Wo);
        t /= Wo;
        const int yo = (int)(t % Ho);
        const long b = t / Ho;
        const __hip_bfloat16 *p = in + (((b * 2 * Ho + 2 * yo) * Wi + 2 * xo) * C + c * 8);
        float a[8], q[8], r[8], d[8], o[8];
        load_vec<__hip_bfloat16, 8>(p, a);
        load_vec<__hip_bfloat16, 8>(p + C, q);
        load_vec<__hip_bfloat16, 8>(p + Wi * C, r);
        load_vec<__hip_bfloat16, 8>(p + Wi * C + C, d);
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (a[j] + q[j]) + (r[j] + d[j]);
        store_vec<__hip_bfloat16, 8>(out + i * 8, o);
    }
}

extern "C" int xq_sumpool2x2_nhwc_bf16(const void *in, int B, int Ho, int Wo, int C, void *out, xq_stream_t stream) {
    const char *fn = "xq_sumpool2x2_nhwc_bf16";
    if (B < 0 || Ho < 1 || Wo < 1 || C < 8 || C % 8) return xq_set_error(XQ_EINVAL, "%s: bad shape", fn);
    if (B == 0) return XQ_OK;
    if (!in || !out) return xq_set_error(XQ_EINVAL, "%s: null pointer", fn);
    const long total = (long)B * Ho * Wo * (C / 8);
    long blocks = (total + 255) / 256;
    const long cap = (long)num_cus() * 32;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(sumpool2x2_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const __hip_bfloat16 *)in, B, Ho, Wo, C,
                       (__hip_bfloat16 *)out);
    return xq_check_launch(fn);
}

// ---------------------------------------------------------------------------------------------------------------------
// MaxPool2d(kernel 2, stride 2) on NHWC bf16 (the four pools of the VGG16 trunk, lpips.py:118-155 via torchvision's cfg).
// Forward: 4 x 16-byte reads -> one 16-byte write per 8 channels.  Backward: recomputes the arg-max from the saved input
// (first maximum in (0,0),(0,1),(1,0),(1,1) order, as ATen's strict '>' scan) instead of reading an int64 index per
// element (ATen stores 8 bytes of index for every 2-byte output: 4x the traffic of the data itself).
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float bfbits_to_f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

__global__ __launch_bounds__(256) void maxpool2x2_fwd_kernel(const unsigned short *__restrict__ X, int B, int Ho, int Wo, int C,
                                                             unsigned short *__restrict__ Y) {
    const int cv = C / 8;
    const long total = (long)B * Ho * Wo * cv;
    const long Wi = 2L * Wo;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % cv);
        long t = i / cv;
        const int xo = (int)(t % Wo);
        t /= Wo;
        const int yo = (int)(t % Ho);
        const long b = t / Ho;
        const unsigned short *p = X + (((b * 2 * Ho + 2 * yo) * Wi + 2 * xo) * C + c * 8);
        const uint4 v00 = *reinterpret_cast<const uint4 *>(p), v01 = *reinterpret_cast<const uint4 *>(p + C);
        const uint4 v10 = *reinterpret_cast<const uint4 *>(p + Wi * C), v11 = *reinterpret_cast<const uint4 *>(p + Wi * C + C);
        const unsigned short *a = reinterpret_cast<const unsigned short *>(&v00), *bq = reinterpret_cast<const unsigned short *>(&v01);
        const unsigned short *cq = reinterpret_cast<const unsigned short *>(&v10), *d = reinterpret_cast<const unsigned short *>(&v11);
        uint4 o;
        unsigned short *oo = reinterpret_cast<unsigned short *>(&o);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            unsigned short m = a[j];
            float mv = bfbits_to_f(m);
            float f = bfbits_to_f(bq[j]);
            if (f > mv) { mv = f; m = bq[j]; }
            f = bfbits_to_f(cq[j]);
            if (f > mv) { mv = f; m = cq[j]; }
            f = bfbits_to_f(d[j]);
            if (f > mv) { mv = f; m = d[j]; }
            oo[j] = m;
        }
        *reinterpret_cast<uint4 *>(Y + i * 8) = o;
    }
}

__global__ __launch_bounds__(256) void maxpool2x2_bwd_kernel(const unsigned short *__restrict__ X, const unsigned short *__restrict__ G,
                                                             int B, int Ho, int Wo, int C, unsigned short *__restrict__ GX) {
    const int cv = C / 8;
    const long total = (long)B * Ho * Wo * cv;
    const long Wi = 2L * Wo;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % cv);
        long t = i / cv;
        const int xo = (int)(t % Wo);
        t /= Wo;
        const int yo = (int)(t % Ho);
        const long b = t / Ho;
        const long off = ((b * 2 * Ho + 2 * yo) * Wi + 2 * xo) * C + c * 8;
        const unsigned short *p = X + off;
        const uint4 v00 = *reinterpret_cast<const uint4 *>(p), v01 = *reinterpret_cast<const uint4 *>(p + C);
        const uint4 v10 = *reinterpret_cast<const uint4 *>(p + Wi * C), v11 = *reinterpret_cast<const uint4 *>(p + Wi * C + C);
        const uint4 gv = *reinterpret_cast<const uint4 *>(G + i * 8);
        const unsigned short *a = reinterpret_cast<const unsigned short *>(&v00), *bq = reinterpret_cast<const unsigned short *>(&v01);
        const unsigned short *cq = reinterpret_cast<const unsigned short *>(&v10), *d = reinterpret_cast<const unsigned short *>(&v11);
        const unsigned short *gg = reinterpret_cast<const unsigned short *>(&gv);
        uint4 o00, o01, o10, o11;
        unsigned short *q00 = reinterpret_cast<unsigned short *>(&o00), *q01 = reinterpret_cast<unsigned short *>(&o01);
        unsigned short *q10 = reinterpret_cast<unsigned short *>(&o10), *q11 = reinterpret_cast<unsigned short *>(&o11);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int arg = 0;
            float mv = bfbits_to_f(a[j]);
            float f = bfbits_to_f(bq[j]);
            if (f > mv) { mv = f; arg = 1; }
            f = bfbits_to_f(cq[j]);
            if (f > mv) { mv = f; arg = 2; }
            f = bfbits_to_f(d[j]);
            if (f > mv) { mv = f; arg = 3; }
            q00[j] = arg == 0 ? gg[j] : (unsigned short)0;
            q01[j] = arg == 1 ? gg[j] : (unsigned short)0;
            q10[j] = arg == 2 ? gg[j] : (unsigned short)0;
            q11[j] = arg == 3 ? gg[j] : (unsigned short)0;
        }
        unsigned short *gp = GX + off;
        *reinterpret_cast<uint4 *>(gp) = o00;
        *reinterpret_cast<uint4 *>(gp + C) = o01;
        *reinterpret_cast<uint4 *>(gp + Wi * C) = o10;
        *reinterpret_cast<uint4 *>(gp + Wi * C + C) = o11;
    }
}

static int pool_check(const char *fn, int B, int Ho, int Wo, int C) {
    if (B < 0 || Ho < 1 || Wo < 1 || C < 8 || C % 8 != 0)
        return xq_set_error(XQ_EINVAL, "%s: needs C %% 8 == 0 and a non-empty output (Wo=%ld, C=%ld)", fn, (long)Wo, (long)C);
    return XQ_OK;
}

extern "C" int xq_maxpool2x2_nhwc_bf16_forward(const void *X, int B, int Ho, int Wo, int C, void *Y, xq_stream_t stream) {
    const char *fn = "xq_maxpool2x2_nhwc_bf16_forward";
    if (int rc = pool_check(fn, B, Ho, Wo, C)) return rc;
    if (B == 0) return XQ_OK;
    if (!X || !Y) return xq_set_error(XQ_EINVAL, "%s: null pointer", fn);
    const long total = (long)B * Ho * Wo * (C / 8);
    long blocks = (total + 255) / 256;
    const long cap = (long)num_cus() * 32;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(maxpool2x2_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const unsigned short *)X, B, Ho, Wo, C,
                       (unsigned short *)Y);
    return xq_check_launch(fn);
}

extern "C" int xq_maxpool2x2_nhwc_bf16_backward(const void *X, const void *G, int B, int Ho, int Wo, int C, void *GX, xq_stream_t stream) {
    const char *fn = "xq_maxpool2x2_nhwc_bf16_backward";
    if (int rc = pool_check(fn, B, Ho, Wo, C)) return rc;
    if (B == 0) return XQ_OK;
    if (!X || !G || !GX) return xq_set_error(XQ_EINVAL, "%s: null pointer", fn);
    const long total = (long)B * Ho * Wo * (C / 8);
    long blocks = (total + 255) / 256;
    const long cap = (long)num_cus() * 32;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(maxpool2x2_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const unsigned short *)X,
                       (const unsigned short *)G, B, Ho, Wo, C, (unsigned short *)GX);
    return xq_check_launch(fn);
}
