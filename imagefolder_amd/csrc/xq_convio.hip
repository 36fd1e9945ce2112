// xq_convio.hip — the two 3-channel 3x3 convolutions at the ends of the CNN tokenizer (gfx950), bf16 arithmetic / fp32 accumulation.
//
//   conv_in  (xqgan_model.py:495: Conv2d(3, 128, 3, 1, 1) on the image)            K = 27:  6.9 kFLOP and 262 B per pixel
//   conv_out (xqgan_model.py:584: Conv2d(128, 3, 3, 1, 1) producing the pixels)    N = 3:   the same, mirrored
// Both are HBM-bound (SURVEY.md §8d names conv_in as the layer to report GB/s on): a 128-channel NHWC activation row is read
// or written once per pixel.  Rounds 1-3 assumed the matrix cores had nothing to contribute at 3 channels; that holds for conv_out (N = 3
// output columns: > 90 % of a tile is padding) but not for conv_in, where the 3 channels are on the K side: 27 taps pad to K = 32.
//   conv3x3_from3_mfma_kernel : planar 3-channel input [B][3][H][W] (fp32 image or bf16 gradient) -> NHWC bf16 [B][H][W][64 | 128] on the
//                          matrix cores, K = 27 padded to 32 (round 4; before: one thread per pixel, 27 x Cout v_fmac, VALU-bound);
//                          conv_in forward, VGG conv1_1, and the data gradient of conv_out (same kernel on the rotated weights);
//   conv3x3_to3_kernel   : NHWC bf16 [B][H][W][C] -> planar bf16 [B][3][H][W]: 9 x C/2 v_dot2c_f32_bf16 per output channel;
//                          conv_out forward;
//   im2col27_kernel      : the 27 (+5 zero) taps of every pixel as a [pixels][32] bf16 matrix: the weight gradient of conv_in is then
//                          the split-K TN GEMM  g[pixels][128]^T . cols[pixels][32]  (xq_gemm_bf16_tn);
//   conv3x3_to3_wgrad_kernel : weight gradient of conv_out: thread = (tap, 8-channel chunk), loops over the pixels of 16 image
//                          rows, 24 fp32 accumulators, per-block partials summed by the caller (deterministic).
#include "xq_common.hpp"
#include "xq_internal.hpp"
#include "../../include/xq_ops.h"

#include <hip/hip_bf16.h>
#include <cstdlib>

using namespace xq;

namespace {

typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float round_bf16(float v) { return __bfloat162float(__float2bfloat16(v)); }
__device__ __forceinline__ float ld_in(const float *p) { return round_bf16(*p); }          // autocast casts the image to bf16
__device__ __forceinline__ float ld_in(const __hip_bfloat16 *p) { return __bfloat162float(*p); }
__device__ __forceinline__ unsigned pack2(float a, float b) {
    const f2 v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf2));
}


// conv3x3_from3 on the matrix cores (round 4; rounds 1-3: one thread per pixel, 27 x Cout v_fmac with SGPR weights — VALU-bound at 1.6-1.8 TB/s,
// profiles/r04_conv_from3.txt).  Wk: fp32 [27][C3_OUT], k = (ky * 3 + kx) * 3 + ci (values already rounded to bf16); bias fp32 [C3_OUT] or null.
// K = 27 padded to 32 = two k-steps of v_mfma_f32_32x32x16_bf16: 16 % padding, not the
// "> 90 %" the header feared for a 3-channel tile — the padding is along K, the 32 x 32 output tile is 32 pixels x 32 output channels, all
// live.  One wave = 32 consecutive pixels x all C3_OUT channels per tile, persistent over tiles: lane (li, hh) gathers the 16 taps
// k = 16 s + 8 hh + j of pixel li straight from the planar image (consecutive lanes = consecutive pixels: coalesced; every input value
// is fetched 9 times, from L1), rounds them to bf16 as autocast does, and the weights sit in registers as bf16 fragments for the life of the
// wave.  27 fp32 FMAs per output element on the VALU become 2 MFMAs per 32 x 32 outputs; what is
// left is the 2 * C3_OUT bytes per pixel of the store (16 bytes per lane through v_permlane32_swap, as conv3x3_c64_kernel).
// Accumulation starts from the bias, k ascending inside the MFMA: fp32 sums of the same 27 exact bf16 x bf16 products as the VALU kernel, in
// another order (results agree with the VALU kernel's to fp32 rounding, i.e. an occasional bf16 ulp).
typedef __bf16 c3_bf16x8 __attribute__((ext_vector_type(8)));
typedef float c3_f32x16 __attribute__((ext_vector_type(16)));

template <typename TIN, int C3_OUT>
__global__ __launch_bounds__(256) void conv3x3_from3_mfma_kernel(const TIN *__restrict__ X, const float *__restrict__ Wk, const float *__restrict__ bias,
                                                                 int B, int H, int W, int relu, char *__restrict__ Y) {
    constexpr int NB = C3_OUT / 32;
    const int lane = threadIdx.x & 63, li = lane & 31, hh = lane >> 5;
    const unsigned total = (unsigned)B * (unsigned)H * (unsigned)W;          // host checks < 2^31
    const unsigned ntiles = (total + 31u) / 32u;
    // this lane's 16 taps: plane / row / column displacement of k = 16 s + 8 hh + j; k >= 27 gets row code 3 (never valid)
    int koff[2][8], kyx[2][8];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 16 * s2 + 8 * hh + j;
            const int tap = (k * 11) >> 5, ci = k - 3 * tap;                 // k / 3 for k < 32
            const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;
            koff[s2][j] = k < 27 ? (ci * H + (ky - 1)) * W + (kx - 1) : 0;
            kyx[s2][j] = k < 27 ? (ky | (kx << 2)) : 3;
        }
    c3_bf16x8 wf[NB][2];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = 16 * s2 + 8 * hh + j;
                wf[nb][s2][j] = (__bf16)(k < 27 ? Wk[k * C3_OUT + 32 * nb + li] : 0.0f);
            }
    __shared__ __attribute__((aligned(16))) float sbias[C3_OUT];      // 16 LDS reads per tile instead of 4 * NB more registers per lane
    if (threadIdx.x < C3_OUT) sbias[threadIdx.x] = bias ? bias[threadIdx.x] : 0.0f;
    __syncthreads();
    const unsigned HW = (unsigned)H * (unsigned)W;
    // tile t -> wave t mod (waves of the grid): at any moment the chip's waves store to ADJACENT 4 KiB runs of the output, spread over all HBM
    // channels.  (Measured and dropped: one contiguous run of tiles per workgroup for L1 reuse of the image rows between vertically adjacent
    // tiles — every workgroup then sits at the same offset of its 1 MiB run and the stores camp on a few channels: 170 -> 225 us on the
    // bf16-input shape, profiles/r04_conv_from3.txt.)
    const unsigned nwaves = gridDim.x * 4u;
    for (unsigned t = blockIdx.x * 4u + (threadIdx.x >> 6); t < ntiles; t += nwaves) {
        const unsigned p = t * 32u + li;
        const bool live = p < total;
        const unsigned pp = live ? p : total - 1u;
        const unsigned b = pp / HW, rem = pp - b * HW;
        const int y = (int)(rem / (unsigned)W), x = (int)(rem - (unsigned)y * (unsigned)W);
        const unsigned rowbits = (y > 0 ? 1u : 0u) | 2u | (y + 1 < H ? 4u : 0u);
        const unsigned colbits = (x > 0 ? 1u : 0u) | 2u | (x + 1 < W ? 4u : 0u);
        const TIN *xp = X + ((long)b * 3 * H + y) * W + x;
        c3_bf16x8 af[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool ok = ((rowbits >> (kyx[s2][j] & 3)) & (colbits >> (kyx[s2][j] >> 2)) & 1u) != 0u;
                af[s2][j] = (__bf16)(ok ? (float)xp[ok ? koff[s2][j] : 0] : 0.0f);
            }
        const bool ok_store = live;
        char *yp = Y + (long)pp * (2 * C3_OUT) + 16 * hh;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            c3_f32x16 acc;      // starts from the bias
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 b4 = *reinterpret_cast<const float4 *>(sbias + 32 * nb + 8 * q + 4 * hh);
                acc[4 * q + 0] = b4.x; acc[4 * q + 1] = b4.y; acc[4 * q + 2] = b4.z; acc[4 * q + 3] = b4.w;
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nb][0], af[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[nb][1], af[1], acc, 0, 0, 0);
            uint2 pk[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float e0 = acc[4 * q + 0], e1 = acc[4 * q + 1], e2 = acc[4 * q + 2], e3 = acc[4 * q + 3];
                if (relu) { e0 = fmaxf(e0, 0.f); e1 = fmaxf(e1, 0.f); e2 = fmaxf(e2, 0.f); e3 = fmaxf(e3, 0.f); }
                pk[q].x = pack2(e0, e1);
                pk[q].y = pack2(e2, e3);
            }
            // lanes 0-31 hold the channels 8 q + 0..3 of their pixel, lanes 32-63 the channels 8 q + 4..7: after the swap a lane of the lower
            // half owns the 8 channels of an even q, its partner those of the odd q (cdna_hip_programming.md T21)
#pragma unroll
            for (int q = 0; q < 4; q += 2) {
                auto rx = __builtin_amdgcn_permlane32_swap(pk[q].x, pk[q + 1].x, false, false);
                auto ry = __builtin_amdgcn_permlane32_swap(pk[q].y, pk[q + 1].y, false, false);
                if (ok_store) *reinterpret_cast<uint4 *>(yp + (32 * nb + 8 * q) * 2) = make_uint4(rx[0], ry[0], rx[1], ry[1]);
            }
        }
    }
}

// Wq: packed bf16 pairs [3][9][C / 2] as uint32; bias fp32 [3] or null; Y planar bf16 [B][3][H][W].
// LPP = C / 8 lanes share a pixel (lane = 8-channel chunk): every wave instruction reads whole contiguous 2 C-byte pixel rows
// (a lane-per-pixel version pulled 8x the useful bytes through L1: 3.5 ms for 32 images at 128 channels).  Each lane keeps its
// 3 x 9 x 4 weight words in registers and walks over pixels; the 3 partial sums are folded over the chunk lanes with xor shuffles.
template <int LPP>
__global__ __launch_bounds__(256) void conv3x3_to3_kernel(const __hip_bfloat16 *__restrict__ X, const unsigned *__restrict__ Wq,
                                                          const float *__restrict__ bias, int B, int H, int W,
                                                          __hip_bfloat16 *__restrict__ Y) {
    constexpr int C = LPP * 8, C2 = C / 2, PPB = 256 / LPP;      // pixels per block and pass
    const int chunk = threadIdx.x % LPP, pl = threadIdx.x / LPP;
    unsigned w[3][9][4];
#pragma unroll
    for (int co = 0; co < 3; ++co)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int q = 0; q < 4; ++q) w[co][tap][q] = Wq[(co * 9 + tap) * C2 + chunk * 4 + q];
    const float b0 = bias ? bias[0] : 0.0f, b1 = bias ? bias[1] : 0.0f, b2 = bias ? bias[2] : 0.0f;
    const long total = (long)B * H * W, plane = (long)H * W;
    for (long p0 = (long)blockIdx.x * PPB; p0 < total; p0 += (long)gridDim.x * PPB) {
        const long p = p0 + pl;
        const bool live = p < total;
        const long pp = live ? p : total - 1;
        const int x = (int)(pp % W), y = (int)((pp / W) % H);
        const long b = pp / plane;
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
            const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
            uint4 v = *reinterpret_cast<const uint4 *>(X + ((b * H + (ok ? yy : y)) * (long)W + (ok ? xx : x)) * C + chunk * 8);
            if (!ok) v = make_uint4(0u, 0u, 0u, 0u);
            const unsigned e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bf2 xv = __builtin_bit_cast(bf2, e[q]);
                a0 = __builtin_amdgcn_fdot2_f32_bf16(xv, __builtin_bit_cast(bf2, w[0][tap][q]), a0, false);
                a1 = __builtin_amdgcn_fdot2_f32_bf16(xv, __builtin_bit_cast(bf2, w[1][tap][q]), a1, false);
                a2 = __builtin_amdgcn_fdot2_f32_bf16(xv, __builtin_bit_cast(bf2, w[2][tap][q]), a2, false);
            }
        }
#pragma unroll
        for (int o = 1; o < LPP; o <<= 1) {
            a0 += __shfl_xor(a0, o);
            a1 += __shfl_xor(a1, o);
            a2 += __shfl_xor(a2, o);
        }
        if (live && chunk == 0) {
            const long o = b * 3 * plane + (long)y * W + x;
            Y[o] = __float2bfloat16(a0 + b0);
            Y[o + plane] = __float2bfloat16(a1 + b1);
            Y[o + 2 * plane] = __float2bfloat16(a2 + b2);
        }
    }
}

// Round 6: the same convolution on 2-D tiles with the input halo in LDS.  The kernel above re-reads every input pixel nine times through L1 / L2
// (its 32-pixel row segments share nothing between grid-stride passes): 9.7 GB of L2 -> L1 traffic for the 1.07 GB gradient of VGG conv1_1, 855 us
// against 0.21 ms of HBM traffic.  Here a block owns TH x 32 output pixels: the (TH + 2) x 34 halo goes global -> registers -> LDS once (16-byte
// chunks, out-of-image pixels as zeros), then the LPP chunk lanes of a pixel read their nine taps from LDS.  Same arithmetic per lane, same fold
// over the chunk lanes, same rounding: bit-identical outputs.  Pixel slots of C * 2 + 16 bytes (the pad keeps the 16-byte reads of neighbouring
// pixels on different banks).
template <int LPP, int TH>
__global__ __launch_bounds__(256) void conv3x3_to3_tiled_kernel(const __hip_bfloat16 *__restrict__ X, const unsigned *__restrict__ Wq,
                                                                const float *__restrict__ bias, int B, int H, int W, int tiles_y, int tiles_x,
                                                                long ntiles, __hip_bfloat16 *__restrict__ Y) {
    constexpr int C = LPP * 8, C2 = C / 2, TW = 32, HW = TW + 2, SLOTS = (TH + 2) * HW, PITCH = C * 2 + 16, PPP = 256 / LPP;   // pixels per pass
    extern __shared__ __attribute__((aligned(16))) char to3_smem[];
    const int chunk = threadIdx.x % LPP, pl = threadIdx.x / LPP;
    unsigned w[3][9][4];
#pragma unroll
    for (int co = 0; co < 3; ++co)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int q = 0; q < 4; ++q) w[co][tap][q] = Wq[(co * 9 + tap) * C2 + chunk * 4 + q];
    const float b0 = bias ? bias[0] : 0.0f, b1 = bias ? bias[1] : 0.0f, b2 = bias ? bias[2] : 0.0f;
    const long plane = (long)H * W;
    for (long t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int tx = (int)(t % tiles_x), ty = (int)((t / tiles_x) % tiles_y);
        const long b = t / ((long)tiles_x * tiles_y);
        const int y0 = ty * TH, x0 = tx * TW;
        __syncthreads();                                   // the previous tile's reads are done
        for (int i = threadIdx.x; i < SLOTS * LPP; i += 256) {
            const int slot = i / LPP, ch = i % LPP;
            const int yy = y0 + slot / HW - 1, xx = x0 + slot % HW - 1;
            const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (ok) v = *reinterpret_cast<const uint4 *>(X + ((b * H + yy) * (long)W + xx) * C + ch * 8);
            *reinterpret_cast<uint4 *>(to3_smem + slot * PITCH + ch * 16) = v;
        }
        __syncthreads();
#pragma unroll 1
        for (int pass = 0; pass < TH * TW / PPP; ++pass) {
            const int idx = pass * PPP + pl, py = idx / TW, px = idx % TW;
            float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const uint4 v = *reinterpret_cast<const uint4 *>(to3_smem + ((py + tap / 3) * HW + px + tap % 3) * PITCH + chunk * 16);
                const unsigned e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const bf2 xv = __builtin_bit_cast(bf2, e[q]);
                    a0 = __builtin_amdgcn_fdot2_f32_bf16(xv, __builtin_bit_cast(bf2, w[0][tap][q]), a0, false);
                    a1 = __builtin_amdgcn_fdot2_f32_bf16(xv, __builtin_bit_cast(bf2, w[1][tap][q]), a1, false);
                    a2 = __builtin_amdgcn_fdot2_f32_bf16(xv, __builtin_bit_cast(bf2, w[2][tap][q]), a2, false);
                }
            }
#pragma unroll
            for (int o = 1; o < LPP; o <<= 1) {
                a0 += __shfl_xor(a0, o);
                a1 += __shfl_xor(a1, o);
                a2 += __shfl_xor(a2, o);
            }
            const int y = y0 + py, x = x0 + px;
            if (chunk == 0 && y < H && x < W) {
                const long o = b * 3 * plane + (long)y * W + x;
                Y[o] = __float2bfloat16(a0 + b0);
                Y[o + plane] = __float2bfloat16(a1 + b1);
                Y[o + 2 * plane] = __float2bfloat16(a2 + b2);
            }
        }
    }
}

// cols[p][k] (bf16, k = (ky*3+kx)*3+ci for k < 27, zero for 27..31)
template <typename TIN>
__global__ __launch_bounds__(256) void im2col27_kernel(const TIN *__restrict__ X, int B, int H, int W, __hip_bfloat16 *__restrict__ cols) {
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)B * H * W;
    if (p >= total) return;
    const int x = (int)(p % W), y = (int)((p / W) % H);
    const long b = p / ((long)W * H);
    float in[32];
#pragma unroll
    for (int k = 27; k < 32; ++k) in[k] = 0.0f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int yy = y + ky - 1, xx = x + kx - 1;
            const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
#pragma unroll
            for (int ci = 0; ci < 3; ++ci)
                in[(ky * 3 + kx) * 3 + ci] = ok ? ld_in(X + ((b * 3 + ci) * H + yy) * (long)W + xx) : 0.0f;
        }
    uint4 *out = reinterpret_cast<uint4 *>(cols + p * 32);
#pragma unroll
    for (int q = 0; q < 4; ++q)
        out[q] = make_uint4(pack2(in[8 * q + 0], in[8 * q + 1]), pack2(in[8 * q + 2], in[8 * q + 3]), pack2(in[8 * q + 4], in[8 * q + 5]),
                            pack2(in[8 * q + 6], in[8 * q + 7]));
}

// part[block][co][tap][c] += sum over this block's rows of g[b][co][y][x] * X[b][y+ky-1][x+kx-1][c]
constexpr int WG3_ROWS = 16;
template <typename TG>
__global__ __launch_bounds__(288) void conv3x3_to3_wgrad_kernel(const __hip_bfloat16 *__restrict__ X, const TG *__restrict__ G, int B, int H, int W, int C,
                                                                float *__restrict__ part) {
    const int t = threadIdx.x, r = t % 144, half = t / 144;
    const int tap = r / 16;
    const int ky = tap / 3, kx = tap - 3 * ky;
    const long row0 = (long)blockIdx.x * WG3_ROWS;
    const long rows = (long)B * H;
    const long plane = (long)H * W;
    for (int cc = (r % 16) * 8; cc < C; cc += 128) {           // 8-channel chunk(s) of this thread
        float acc[3][8];
#pragma unroll
        for (int co = 0; co < 3; ++co)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[co][j] = 0.0f;
        for (long rr = row0; rr < row0 + WG3_ROWS && rr < rows; ++rr) {
            const long b = rr / H;
            const int y = (int)(rr - b * H);
            const int yy = y + ky - 1;
            if (yy < 0 || yy >= H) continue;
            const __hip_bfloat16 *xrow = X + ((b * H + yy) * (long)W) * C + cc;
            const TG *g0 = G + b * 3 * plane + (long)y * W;
            const int xb = half * (W / 2), xe = half ? W : W / 2;
            for (int x = xb; x < xe; ++x) {
                const int xx = x + kx - 1;
                if (xx < 0 || xx >= W) continue;
                const uint4 v = *reinterpret_cast<const uint4 *>(xrow + (long)xx * C);
                const float ga = ld_in(g0 + x), gb = ld_in(g0 + plane + x), gc = ld_in(g0 + 2 * plane + x);
                const unsigned e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float lo = __uint_as_float(e[q] << 16), hi = __uint_as_float(e[q] & 0xffff0000u);
                    acc[0][2 * q] = __builtin_fmaf(ga, lo, acc[0][2 * q]); acc[0][2 * q + 1] = __builtin_fmaf(ga, hi, acc[0][2 * q + 1]);
                    acc[1][2 * q] = __builtin_fmaf(gb, lo, acc[1][2 * q]); acc[1][2 * q + 1] = __builtin_fmaf(gb, hi, acc[1][2 * q + 1]);
                    acc[2][2 * q] = __builtin_fmaf(gc, lo, acc[2][2 * q]); acc[2][2 * q + 1] = __builtin_fmaf(gc, hi, acc[2][2 * q + 1]);
                }
            }
        }
        // the two halves of the row share (tap, chunk): combine through LDS, fixed order
        __shared__ float red[144][24];
        if (half == 1) {
#pragma unroll
            for (int co = 0; co < 3; ++co)
#pragma unroll
                for (int j = 0; j < 8; ++j) red[r][co * 8 + j] = acc[co][j];
        }
        __syncthreads();
        if (half == 0) {
            float *o = part + (size_t)blockIdx.x * 27 * C;
#pragma unroll
            for (int co = 0; co < 3; ++co)
#pragma unroll
                for (int j = 0; j < 8; ++j) o[(co * 9 + tap) * C + cc + j] = acc[co][j] + red[r][co * 8 + j];
        }
        __syncthreads();
    }
}

// row softmax of the AttnBlock scores (xqgan_model.py:652-653): s = bf16(S * scale) (the bf16 product of the bmm output with the
// Python scalar), p = softmax(s) in fp32 (autocast runs softmax in fp32); p32 is kept for the backward, p16 feeds the second bmm
__global__ __launch_bounds__(256) void row_softmax_fwd_kernel(const __hip_bfloat16 *__restrict__ S, long rows, int N, float scale,
                                                              float *__restrict__ P32, __hip_bfloat16 *__restrict__ P16) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    const __hip_bfloat16 *s = S + row * N;
    float v[16];
    float mx = -__builtin_inff();
    const int per = N / 64;
    for (int j = 0; j < per; ++j) {
        v[j] = round_bf16(__bfloat162float(s[j * 64 + lane]) * scale);
        mx = fmaxf(mx, v[j]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.0f;
    for (int j = 0; j < per; ++j) { v[j] = expf(v[j] - mx); sum += v[j]; }
    sum = wave_sum(sum);
    const float inv = 1.0f / sum;
    for (int j = 0; j < per; ++j) {
        const float p = v[j] * inv;
        P32[row * N + j * 64 + lane] = p;
        P16[row * N + j * 64 + lane] = __float2bfloat16(p);
    }
}

// dS = bf16(bf16(p * (dP - sum_j p_j dP_j)) * scale)
__global__ __launch_bounds__(256) void row_softmax_bwd_kernel(const float *__restrict__ P32, const __hip_bfloat16 *__restrict__ dP, long rows, int N,
                                                              float scale, __hip_bfloat16 *__restrict__ dS) {
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= rows) return;
    float p[16], g[16];
    float dot = 0.0f;
    const int per = N / 64;
    for (int j = 0; j < per; ++j) {
        p[j] = P32[row * N + j * 64 + lane];
        g[j] = __bfloat162float(dP[row * N + j * 64 + lane]);
        dot = __builtin_fmaf(p[j], g[j], dot);
    }
    dot = wave_sum(dot);
    for (int j = 0; j < per; ++j) dS[row * N + j * 64 + lane] = __float2bfloat16(round_bf16(p[j] * (g[j] - dot)) * scale);
}

}  // namespace

extern "C" int xq_row_softmax_forward(const void *S, int64_t rows, int N, float scale, float *P32, void *P16, xq_stream_t stream) {
    const char *fn = "xq_row_softmax_forward";
    if (rows < 0 || N < 64 || N % 64 || N > 1024) return xq_set_error(XQ_EINVAL, "%s: N must be a multiple of 64, <= 1024", fn);
    if (rows == 0) return XQ_OK;
    if (!S || !P32 || !P16) return xq_set_error(XQ_EINVAL, "%s: null pointer", fn);
    hipLaunchKernelGGL(row_softmax_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const __hip_bfloat16 *)S, (long)rows, N,
                       scale, P32, (__hip_bfloat16 *)P16);
    return xq_check_launch(fn);
}

extern "C" int xq_row_softmax_backward(const float *P32, const void *dP, int64_t rows, int N, float scale, void *dS, xq_stream_t stream) {
    const char *fn = "xq_row_softmax_backward";
    if (rows < 0 || N < 64 || N % 64 || N > 1024) return xq_set_error(XQ_EINVAL, "%s: N must be a multiple of 64, <= 1024", fn);
    if (rows == 0) return XQ_OK;
    if (!P32 || !dP || !dS) return xq_set_error(XQ_EINVAL, "%s: null pointer", fn);
    hipLaunchKernelGGL(row_softmax_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, P32, (const __hip_bfloat16 *)dP, (long)rows,
                       N, scale, (__hip_bfloat16 *)dS);
    return xq_check_launch(fn);
}

extern "C" int xq_conv3x3_from3_forward(const void *x_planar, int x_is_bf16, const float *w_kc, const float *bias, int B, int H, int W, int Cout,
                                        int relu, void *y_nhwc, xq_stream_t stream) {
    const char *fn = "xq_conv3x3_from3_forward";
    if (B < 0 || H < 1 || W < 1 || (Cout != 64 && Cout != 128)) return xq_set_error(XQ_EINVAL, "%s: bad shape (Cout 64 or 128)", fn);
    if (B == 0) return XQ_OK;
    if (!x_planar || !w_kc || !y_nhwc) return xq_set_error(XQ_EINVAL, "%s: null pointer", fn);
    const long total = (long)B * H * W;
    if (total * 3 >= 0x7fffffffL) return xq_set_error(XQ_EINVAL, "%s: image batch too large for 32-bit pixel indices", fn);
    hipStream_t s = (hipStream_t)stream;
    const long tiles = (total + 31) / 32;
    long blocks = (tiles + 3) / 4;
    const long cap = (long)num_cus() * (Cout == 128 ? 3 : 4);      // resident workgroups per CU at 137 / 121 VGPRs; each wave walks its tiles
    if (blocks > cap) blocks = cap;
    const int pslot = xq::prof_begin(XQ_PROF_CONV_FROM3, (double)total * (3.0 * (x_is_bf16 ? 2.0 : 4.0) + 2.0 * Cout), s);
#define FROM3M(T, CO) hipLaunchKernelGGL((conv3x3_from3_mfma_kernel<T, CO>), dim3((unsigned)blocks), dim3(256), 0, s, (const T *)x_planar, w_kc, bias, B, H, W, relu, (char *)y_nhwc)
    if (x_is_bf16) { if (Cout == 128) FROM3M(__hip_bfloat16, 128); else FROM3M(__hip_bfloat16, 64); }
    else { if (Cout == 128) FROM3M(float, 128); else FROM3M(float, 64); }
#undef FROM3M
    xq::prof_end(pslot, s);
    return xq_check_launch(fn);
}

extern "C" int xq_conv3x3_to3_forward(const void *x_nhwc, const void *w_pairs, const float *bias, int B, int H, int W, int C, void *y_planar,
                                      xq_stream_t stream) {
    const char *fn = "xq_conv3x3_to3_forward";
    if (B < 0 || H < 1 || W < 1 || (C != 64 && C != 128)) return xq_set_error(XQ_EINVAL, "%s: bad shape (C = 64 or 128)", fn);
    if (B == 0) return XQ_OK;
    if (!x_nhwc || !w_pairs || !y_planar) return xq_set_error(XQ_EINVAL, "%s: null pointer", fn);
    const long total = (long)B * H * W;
    hipStream_t s = (hipStream_t)stream;
    // (measured and dropped in round 4: the same convolution on the matrix cores with the three output channels as rows 0 .. 2 of a 32-row
    // weight operand and per-lane 16-byte gathers of the pixels — 1040 us against this kernel's 874 us at 64 channels, 745 against 425 at 128:
    // a lane-per-pixel gather touches 64 cache lines per load instruction where the 8-lanes-per-pixel form below touches 8,
    // profiles/r04_conv_to3_mfma_rejected.txt)
    static const int tiled = [] { const char *e = getenv("XQ_TO3_TILED"); return e ? atoi(e) : 1; }();      // 1: 2-D tiles with the halo in LDS at 64 channels (round 6); 0: row segments
    if (tiled && C == 64) {
        // 8 x 32 output pixels per tile: 10 x 34 halo slots of 144 B = 47.8 KiB, three blocks per CU.  128 x 65 536 pixels x 64 channels (the data
        // gradient of VGG conv1_1): 0.868 -> 0.572 ms, bit-identical (tools/bench_to3.py, profiles/r06_conv_to3_tiled.txt).  At 128 channels the form
        // with 4-row tiles (54 KiB) measured 0.496 against 0.456 ms for the row-segment kernel: kept on the latter.
        constexpr int TH = 8;
        const int tiles_y = (H + TH - 1) / TH, tiles_x = (W + 31) / 32;
        const long ntiles = (long)B * tiles_y * tiles_x;
        const int lds = (TH + 2) * 34 * (64 * 2 + 16);
        long blocks = ntiles;
        const long cap = (long)num_cus() * 3;
        if (blocks > cap) blocks = cap;
        auto k = conv3x3_to3_tiled_kernel<8, TH>;
        static unsigned long long devs = 0;
        if (first_call_on_this_device(&devs) && hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
            return xq_set_error(XQ_ELAUNCH, "%s: cannot reserve %d bytes of LDS", fn, lds);
        hipLaunchKernelGGL(k, dim3((unsigned)blocks), dim3(256), lds, s, (const __hip_bfloat16 *)x_nhwc, (const unsigned *)w_pairs, bias, B, H, W, tiles_y,
                           tiles_x, ntiles, (__hip_bfloat16 *)y_planar);
        return xq_check_launch(fn);
    }
    const int ppb = 256 / (C / 8);
    long blocks = (total + ppb - 1) / ppb;
    const long cap = (long)num_cus() * 16;
    if (blocks > cap) blocks = cap;
    if (C == 128) hipLaunchKernelGGL((conv3x3_to3_kernel<16>), dim3((unsigned)blocks), dim3(256), 0, s, (const __hip_bfloat16 *)x_nhwc, (const unsigned *)w_pairs, bias, B, H, W, (__hip_bfloat16 *)y_planar);
    else hipLaunchKernelGGL((conv3x3_to3_kernel<8>), dim3((unsigned)blocks), dim3(256), 0, s, (const __hip_bfloat16 *)x_nhwc, (const unsigned *)w_pairs, bias, B, H, W, (__hip_bfloat16 *)y_planar);
    return xq_check_launch(fn);
}

extern "C" int xq_im2col27(const void *x_planar, int x_is_bf16, int B, int H, int W, void *cols, xq_stream_t stream) {
    const char *fn = "xq_im2col27";
    if (B < 0 || H < 1 || W < 1) return xq_set_error(XQ_EINVAL, "%s: bad shape", fn);
    if (B == 0) return XQ_OK;
    if (!x_planar || !cols) return xq_set_error(XQ_EINVAL, "%s: null pointer", fn);
    const long total = (long)B * H * W;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    hipStream_t s = (hipStream_t)stream;
    if (x_is_bf16) hipLaunchKernelGGL((im2col27_kernel<__hip_bfloat16>), dim3(blocks), dim3(256), 0, s, (const __hip_bfloat16 *)x_planar, B, H, W, (__hip_bfloat16 *)cols);
    else hipLaunchKernelGGL((im2col27_kernel<float>), dim3(blocks), dim3(256), 0, s, (const float *)x_planar, B, H, W, (__hip_bfloat16 *)cols);
    return xq_check_launch(fn);
}

extern "C" int xq_conv3x3_to3_wgrad_blocks(int B, int H) { return (int)(((long)B * H + WG3_ROWS - 1) / WG3_ROWS); }

extern "C" int xq_conv3x3_to3_wgrad(const void *x_nhwc, const void *g_planar, int g_is_bf16, int B, int H, int W, int C, float *partials,
                                    xq_stream_t stream) {
    const char *fn = "xq_conv3x3_to3_wgrad";
    if (B < 0 || H < 1 || W < 2 || W % 2 || C < 8 || C % 128) return xq_set_error(XQ_EINVAL, "%s: bad shape (even W, C %% 128 == 0)", fn);
    if (B == 0) return XQ_OK;
    if (!x_nhwc || !g_planar || !partials) return xq_set_error(XQ_EINVAL, "%s: null pointer", fn);
    const unsigned blocks = (unsigned)xq_conv3x3_to3_wgrad_blocks(B, H);
    hipStream_t s = (hipStream_t)stream;
    if (g_is_bf16) hipLaunchKernelGGL((conv3x3_to3_wgrad_kernel<__hip_bfloat16>), dim3(blocks), dim3(288), 0, s, (const __hip_bfloat16 *)x_nhwc, (const __hip_bfloat16 *)g_planar, B, H, W, C, partials);
    else hipLaunchKernelGGL((conv3x3_to3_wgrad_kernel<float>), dim3(blocks), dim3(288), 0, s, (const __hip_bfloat16 *)x_nhwc, (const float *)g_planar, B, H, W, C, partials);
    return xq_check_launch(fn);
}
