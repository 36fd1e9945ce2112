// xq_dense.hip — HBM-bound fused row kernels of the ViT encoder/decoder blocks (gfx950).
//
// Replaces the unfused ATen elementwise/normalisation chain of the reference's timm blocks
// (tokenizer/tokenizer_image/dino_enc/vision_transformer.py:295-339: LayerNorm -> ... -> LayerScale -> DropPath ->
//  residual add; Mlp: fc1 bias -> GELU), which under bf16 autocast runs as ~20 separate kernels per block on an fp32
// residual stream (profiles/r01_train_step_aten_v2_kernel_stats.txt: ~100 ms of a 236 ms step):
//   res_ln_fwd : x_new = x + mask_b * (gamma * y)      (LayerScale :291, DropPath, residual :337-338)
//                a     = LayerNorm(x_new) * w + b      (next norm1/norm2/final norm, eps 1e-6)
//                one pass: reads x (fp32) + y (T), writes x_new (fp32) + a (T) + per-row mean/rstd
//   res_ln_bwd : the transpose of the above in one pass + per-block column partial sums for d_gamma, d_w, d_b and the
//                bias gradient of the Linear that produced y
//   gelu_fwd / gelu_bwd : exact (erf) GELU on the fc1 output (+ column partials for the fc1 bias gradient)
//   colsum_finalize : fixed-order reduction of the column partials into the parameter gradients
// T = activation dtype (bf16 in training, fp32 for the fp32-parity tests); statistics and the residual stream are
// fp32, as they are under the reference's autocast.  One wave per row, 16-byte (or 8-byte) lane accesses laid out so
// that every wave instruction touches one contiguous 1 KB / 512 B segment.
#include "xq_common.hpp"
#include "xq_internal.hpp"
#include "../../include/xq_ops.h"

#include "xq_vec.hpp"
#include <cstdlib>
#include "xq_act.hpp"

using namespace xq;

#ifndef XQ_MAX_ROW_BLOCKS
#define XQ_MAX_ROW_BLOCKS 2048   // 8 blocks = 32 waves per CU: the row kernels are bound by bytes in flight (512: 3.9 TB/s, 2048: 4.4 TB/s backward at 65 664 x 768)
#endif
#ifndef XQ_FWD_BLOCKS_PER_CU
#define XQ_FWD_BLOCKS_PER_CU 8
#endif
static constexpr int ROW_THREADS = 256;  // 4 waves = 4 rows in flight per block
static constexpr int MAX_ROW_BLOCKS = XQ_MAX_ROW_BLOCKS;   // also the number of partial rows the finalize kernel reduces

// element index of (chunk k, lane l, j): k*64*VEC + l*VEC + j  -> each wave instruction reads 64*VEC contiguous elements
template <typename T, int NV, int VEC>
__global__ __launch_bounds__(ROW_THREADS) void res_ln_fwd_kernel(const float *__restrict__ x, const T *__restrict__ y,
                                                                 const float *__restrict__ gamma, const float *__restrict__ mask,
                                                                 long rows, int rows_per_sample, const float *__restrict__ lnw,
                                                                 const float *__restrict__ lnb, float eps, float *__restrict__ x_new,
                                                                 T *__restrict__ a, float *__restrict__ mean_out,
                                                                 float *__restrict__ rstd_out) {
    constexpr int D = NV * VEC * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long r = (long)blockIdx.x * 4 + wave; r < rows; r += (long)gridDim.x * 4) {
        const float m = mask ? mask[r / rows_per_sample] : 1.0f;
        float v[NV][VEC];
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int c = k * 64 * VEC + lane * VEC;
            load_vec<float, VEC>(x + r * D + c, v[k]);
            if (y) {
                float yy[VEC], gg[VEC];
                load_vec<T, VEC>(y + r * D + c, yy);
                if (gamma) load_vec<float, VEC>(gamma + c, gg);
#pragma unroll
                for (int j = 0; j < VEC; ++j) v[k][j] = v[k][j] + m * ((gamma ? gg[j] : 1.0f) * yy[j]);
            }
#pragma unroll
            for (int j = 0; j < VEC; ++j) s += v[k][j];
            if (x_new) store_vec<float, VEC>(x_new + r * D + c, v[k]);
        }
        const float mean = wave_sum(s) * (1.0f / D);
        float q = 0.0f;
#pragma unroll
        for (int k = 0; k < NV; ++k)
#pragma unroll
            for (int j = 0; j < VEC; ++j) { const float d = v[k][j] - mean; q = __builtin_fmaf(d, d, q); }
        const float var = wave_sum(q) * (1.0f / D);
        const float rstd = 1.0f / __builtin_sqrtf(var + eps);
        if (lane == 0) { mean_out[r] = mean; rstd_out[r] = rstd; }
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int c = k * 64 * VEC + lane * VEC;
            float ww[VEC], bb[VEC], o[VEC];
            load_vec<float, VEC>(lnw + c, ww);
            load_vec<float, VEC>(lnb + c, bb);
#pragma unroll
            for (int j = 0; j < VEC; ++j) o[j] = (v[k][j] - mean) * rstd * ww[j] + bb[j];
            store_vec<T, VEC>(a + r * D + c, o);
        }
    }
}

// partials layout: [block][4 quantities][D]: 0 = d_lnw, 1 = d_lnb, 2 = d_gamma, 3 = d_bias(y)
template <typename T, int NV, int VEC>
__global__ __launch_bounds__(ROW_THREADS) void res_ln_bwd_kernel(const T *__restrict__ g_a, const float *__restrict__ g_xnew,
                                                                 const float *__restrict__ x_new, const float *__restrict__ mean,
                                                                 const float *__restrict__ rstd, const float *__restrict__ lnw,
                                                                 const T *__restrict__ y, const float *__restrict__ gamma,
                                                                 const float *__restrict__ mask, long rows, int rows_per_sample,
                                                                 float *__restrict__ g_x, T *__restrict__ g_y,
                                                                 float *__restrict__ partials) {
    constexpr int D = NV * VEC * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float p_w[NV][VEC], p_b[NV][VEC], p_g[NV][VEC], p_y[NV][VEC];
#pragma unroll
    for (int k = 0; k < NV; ++k)
#pragma unroll
        for (int j = 0; j < VEC; ++j) { p_w[k][j] = 0.f; p_b[k][j] = 0.f; p_g[k][j] = 0.f; p_y[k][j] = 0.f; }
    for (long r = (long)blockIdx.x * 4 + wave; r < rows; r += (long)gridDim.x * 4) {
        const float m = mask ? mask[r / rows_per_sample] : 1.0f;
        const float mu = mean[r], rs = rstd[r];
        float xh[NV][VEC], gxh[NV][VEC];
        float c1 = 0.0f, c2 = 0.0f;
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int c = k * 64 * VEC + lane * VEC;
            float xv[VEC], ga[VEC], ww[VEC];
            load_vec<float, VEC>(x_new + r * D + c, xv);
            if (g_a) load_vec<T, VEC>(g_a + r * D + c, ga);
            load_vec<float, VEC>(lnw + c, ww);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const float g = g_a ? ga[j] : 0.0f;
                xh[k][j] = (xv[j] - mu) * rs;
                gxh[k][j] = g * ww[j];
                p_w[k][j] = __builtin_fmaf(g, xh[k][j], p_w[k][j]);
                p_b[k][j] += g;
                c1 += gxh[k][j];
                c2 = __builtin_fmaf(gxh[k][j], xh[k][j], c2);
            }
        }
        c1 = wave_sum(c1) * (1.0f / D);
        c2 = wave_sum(c2) * (1.0f / D);
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int c = k * 64 * VEC + lane * VEC;
            float gt[VEC], gx_in[VEC];
            if (g_xnew) load_vec<float, VEC>(g_xnew + r * D + c, gx_in);
#pragma unroll
            for (int j = 0; j < VEC; ++j) gt[j] = (g_xnew ? gx_in[j] : 0.0f) + rs * (gxh[k][j] - c1 - xh[k][j] * c2);
            store_vec<float, VEC>(g_x + r * D + c, gt);
            if (y) {
                float yy[VEC], gg[VEC], gy[VEC];
                load_vec<T, VEC>(y + r * D + c, yy);
                if (gamma) load_vec<float, VEC>(gamma + c, gg);
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    gy[j] = m * (gamma ? gg[j] : 1.0f) * gt[j];
                    p_g[k][j] = __builtin_fmaf(m * yy[j], gt[j], p_g[k][j]);
                    p_y[k][j] += to_f<T>(from_f<T>(gy[j]));  // the bias gradient sums the stored (rounded) g_y
                }
                store_vec<T, VEC>(g_y + r * D + c, gy);
            }
        }
    }
    // combine the 4 waves of the block, then one partial row per block
    __shared__ float red[4][4][NV * VEC * 64];
#pragma unroll
    for (int k = 0; k < NV; ++k)
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const int c = k * 64 * VEC + lane * VEC + j;
            red[wave][0][c] = p_w[k][j]; red[wave][1][c] = p_b[k][j]; red[wave][2][c] = p_g[k][j]; red[wave][3][c] = p_y[k][j];
        }
    __syncthreads();
    for (int i = threadIdx.x; i < 4 * D; i += ROW_THREADS) {
        const int qn = i / D, c = i - qn * D;
        partials[((size_t)blockIdx.x * 4 + qn) * D + c] = (red[0][qn][c] + red[1][qn][c]) + (red[2][qn][c] + red[3][qn][c]);
    }
}

// Round 6: the same backward with the COLUMNS of a row split over the waves of a block (wave w owns chunk w = 64 * VEC columns) and R rows per
// iteration.  What it removes from the kernel above: 3/4 of the column-partial registers (4 * VEC instead of 4 * NV * VEC per lane: every row a wave
// touches adds into the same columns), the per-row reloads of lnw / gamma (a wave's columns never change: loaded once), the 48 KiB LDS combine of
// the four waves' partials (a wave owns its columns: it writes them straight to the partial row), and the second memory round trip per row (all
// four tensors of the R rows are requested up front).  What it adds: the two row sums c1, c2 are per-chunk partial sums that the NV waves exchange
// through 2 * R floats of LDS each and one __syncthreads per iteration (slots alternate, so one barrier orders both the reads and the reuse).
// 168 -> 64 VGPRs at R = 2 (five blocks of three waves per CU at D = 768).  Same partials layout [block][4][D]; the grid is exactly the resident
// number of blocks.  profiles/r06_res_ln_bwd_cols.txt.
template <typename T, int NV, int VEC, int R, int MINW = 1>
__global__ __launch_bounds__(NV * 64, MINW) void res_ln_bwd_cols_kernel(const T *__restrict__ g_a, const float *__restrict__ g_xnew,
                                                                  const float *__restrict__ x_new, const float *__restrict__ mean,
                                                                  const float *__restrict__ rstd, const float *__restrict__ lnw,
                                                                  const T *__restrict__ y, const float *__restrict__ gamma,
                                                                  const float *__restrict__ mask, long rows, int rows_per_sample,
                                                                  float *__restrict__ g_x, T *__restrict__ g_y,
                                                                  float *__restrict__ partials) {
    constexpr int D = NV * VEC * 64;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = wave * 64 * VEC + lane * VEC;
    __shared__ float xch[2][NV][2 * R];
    float ww[VEC], gg[VEC];
    load_vec<float, VEC>(lnw + c, ww);
#pragma unroll
    for (int j = 0; j < VEC; ++j) gg[j] = 1.0f;
    if (y && gamma) load_vec<float, VEC>(gamma + c, gg);
    float p_w[VEC], p_b[VEC], p_g[VEC], p_y[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { p_w[j] = 0.f; p_b[j] = 0.f; p_g[j] = 0.f; p_y[j] = 0.f; }
    const long groups = (rows + R - 1) / R;
    int slot = 0;
    for (long grp = blockIdx.x; grp < groups; grp += gridDim.x, slot ^= 1) {
        const long r0 = grp * R;
        float xh[R][VEC], gxh[R][VEC], gin[R][VEC], yy[R][VEC], mu[R], rs[R], mk[R];
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const long r = r0 + i;
            const bool ok = r < rows;          // (uniform)
#pragma unroll
            for (int j = 0; j < VEC; ++j) { xh[i][j] = 0.f; gxh[i][j] = 0.f; gin[i][j] = 0.f; yy[i][j] = 0.f; }
            mu[i] = 0.f; rs[i] = 0.f; mk[i] = 0.f;
            if (ok) {
                load_vec<float, VEC>(x_new + r * D + c, xh[i]);
                if (g_a) load_vec<T, VEC>(g_a + r * D + c, gxh[i]);
                if (g_xnew) load_vec<float, VEC>(g_xnew + r * D + c, gin[i]);
                if (y) load_vec<T, VEC>(y + r * D + c, yy[i]);
                mu[i] = mean[r]; rs[i] = rstd[r];
                mk[i] = mask ? mask[r / rows_per_sample] : 1.0f;
            }
        }
        float cs[2 * R];
#pragma unroll
        for (int i = 0; i < R; ++i) {
            float c1 = 0.f, c2 = 0.f;
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const float g = gxh[i][j];
                const float xn = (xh[i][j] - mu[i]) * rs[i];
                const float gw = g * ww[j];
                xh[i][j] = xn;
                gxh[i][j] = gw;
                p_w[j] = __builtin_fmaf(g, xn, p_w[j]);
                p_b[j] += g;
                c1 += gw;
                c2 = __builtin_fmaf(gw, xn, c2);
            }
            cs[2 * i] = wave_sum(c1);
            cs[2 * i + 1] = wave_sum(c2);
        }
        if (NV > 1) {
            if (lane == 0) {
#pragma unroll
                for (int q = 0; q < 2 * R; ++q) xch[slot][wave][q] = cs[q];
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 2 * R; ++q) {
                float t = xch[slot][0][q];
#pragma unroll
                for (int w = 1; w < NV; ++w) t += xch[slot][w][q];
                cs[q] = t;
            }
        }
#pragma unroll
        for (int i = 0; i < R; ++i) {
            const long r = r0 + i;
            if (r < rows) {
                const float c1 = cs[2 * i] * (1.0f / D), c2 = cs[2 * i + 1] * (1.0f / D);
                float gt[VEC];
#pragma unroll
                for (int j = 0; j < VEC; ++j) gt[j] = gin[i][j] + rs[i] * (gxh[i][j] - c1 - xh[i][j] * c2);
                store_vec<float, VEC>(g_x + r * D + c, gt);
                if (y) {
                    float gy[VEC];
#pragma unroll
                    for (int j = 0; j < VEC; ++j) {
                        gy[j] = mk[i] * gg[j] * gt[j];
                        p_g[j] = __builtin_fmaf(mk[i] * yy[i][j], gt[j], p_g[j]);
                        p_y[j] += to_f<T>(from_f<T>(gy[j]));  // the bias gradient sums the stored (rounded) g_y
                    }
                    store_vec<T, VEC>(g_y + r * D + c, gy);
                }
            }
        }
    }
    float *pr = partials + (size_t)blockIdx.x * 4 * D + c;
    store_vec<float, VEC>(pr, p_w);
    store_vec<float, VEC>(pr + D, p_b);
    store_vec<float, VEC>(pr + 2 * D, p_g);
    store_vec<float, VEC>(pr + 3 * D, p_y);
}

// out[q][c] (+)= sum over blocks of partials[block][q][c] in a fixed order.  One 256-thread block per 16 columns:
// 16 row lanes x 16 columns (64-byte segments), each lane strides over the partial rows (up to 32 independent loads in
// flight per thread), then the 16 lanes are combined through LDS in a fixed order.  (The earlier 64-column x 4-lane shape
// left 128 serial loads per thread and only ~48 blocks: 42 us per call, 6 ms per train step.)
static constexpr int FIN_COLS = 16, FIN_LANES = 16;
// block size of the column kernels: one thread per 16-byte column vector, whole waves, at most 1024
static inline int col_threads(int vectors_per_row) {
    int t = (vectors_per_row + 63) / 64 * 64;
    return t < 64 ? 64 : (t > 1024 ? 1024 : t);
}
__global__ __launch_bounds__(256) void colsum_finalize_kernel(const float *__restrict__ partials, int nblocks, int nq, int D,
                                                              float *o0, float *o1, float *o2, float *o3, int accumulate) {
    const int cx = threadIdx.x % FIN_COLS, ry = threadIdx.x / FIN_COLS;
    const int i = blockIdx.x * FIN_COLS + cx;  // flat (q, c) index
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    int qn = 0, c = 0;
    if (i < nq * D) {
        qn = i / D;
        c = i - qn * D;
        const float *p = partials + (size_t)qn * D + c;
        const size_t stride = (size_t)nq * D;
        int b = ry;
        for (; b + 3 * FIN_LANES < nblocks; b += 4 * FIN_LANES) {
            s0 += p[(size_t)b * stride];
            s1 += p[(size_t)(b + FIN_LANES) * stride];
            s2 += p[(size_t)(b + 2 * FIN_LANES) * stride];
            s3 += p[(size_t)(b + 3 * FIN_LANES) * stride];
        }
        for (; b < nblocks; b += FIN_LANES) s0 += p[(size_t)b * stride];
    }
    __shared__ float red[FIN_LANES][FIN_COLS];
    red[ry][cx] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (ry == 0 && i < nq * D) {
        float *out = qn == 0 ? o0 : (qn == 1 ? o1 : (qn == 2 ? o2 : o3));
        if (out) {
            float t = 0.0f;
#pragma unroll
            for (int r = 0; r < FIN_LANES; ++r) t += red[r][cx];
            out[c] = accumulate ? out[c] + t : t;
        }
    }
}

// First level of the column sums when a kernel left many partial rows (2048 blocks x 4 D floats = 25 MB per LayerNorm backward at D = 768):
// grid (width / 64, FIN_CHUNKS) — each block sums one chunk of the rows for 64 columns with 16-byte loads, 256 contiguous bytes per row, in
// a fixed order — into FIN_CHUNKS rows behind the partials (xq_row_partials_blocks reserves them); colsum_finalize_kernel then sums those.
// (The one-level kernel alone: 192 blocks walking 2048 rows in 64-byte pieces, 13 - 20 us per call, 175 calls per train step.)
static constexpr int FIN_CHUNKS = 8, FIN_TWO_LEVEL_FROM = 128;
__global__ __launch_bounds__(256) void colsum_stage_kernel(const float *__restrict__ partials, int nblocks, int width, float *__restrict__ tmp) {
    const int cq = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c4 = (blockIdx.x * 16 + cq) * 4;
    const int per = (nblocks + gridDim.y - 1) / gridDim.y;
    const int r0 = blockIdx.y * per, r1 = min(nblocks, r0 + per);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a, c = a, d = a;
    if (c4 < width) {
        const float *p = partials + c4;
        int r = r0 + rl;
#define FIN_ADD(ACC, R) { const float4 v = *reinterpret_cast<const float4 *>(p + (size_t)(R) * width); ACC.x += v.x; ACC.y += v.y; ACC.z += v.z; ACC.w += v.w; }
        for (; r + 48 < r1; r += 64) { FIN_ADD(a, r) FIN_ADD(b, r + 16) FIN_ADD(c, r + 32) FIN_ADD(d, r + 48) }
        for (; r < r1; r += 16) FIN_ADD(a, r)
#undef FIN_ADD
    }
    __shared__ float4 red[16][16];
    red[rl][cq] = make_float4((a.x + b.x) + (c.x + d.x), (a.y + b.y) + (c.y + d.y), (a.z + b.z) + (c.z + d.z), (a.w + b.w) + (c.w + d.w));
    __syncthreads();
    if (rl == 0 && c4 < width) {
        float4 t = red[0][cq];
#pragma unroll
        for (int k = 1; k < 16; ++k) { const float4 v = red[k][cq]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        *reinterpret_cast<float4 *>(tmp + (size_t)blockIdx.y * width + c4) = t;
    }
}

template <typename T, bool TANH>
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const T *__restrict__ h, long n, T *__restrict__ out) {
    constexpr int VEC = 16 / sizeof(T);
    const long nv = n / VEC;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long)gridDim.x * 256) {
        float v[VEC];
        load_vec<T, VEC>(h + i * VEC, v);
#pragma unroll
        for (int j = 0; j < VEC; ++j) v[j] = gelu_val<TANH>(v[j]);
        store_vec<T, VEC>(out + i * VEC, v);
    }
}

// g_h = g_out * gelu'(h); column partial sums of g_h (fc1 bias gradient): partials[block][H]
template <typename T, bool TANH>
__global__ __launch_bounds__(1024) void gelu_bwd_kernel(const T *__restrict__ g_out, const T *__restrict__ h, long rows, int H,
                                                       T *__restrict__ g_h, float *__restrict__ partials) {
    constexpr int VEC = 16 / sizeof(T);
    // thread t owns columns [t*VEC, t*VEC+VEC) (+ k*blockDim*VEC); rows strided over blocks.  The launcher sizes the block to
    // H/VEC threads (one pass, every thread busy: with 256-thread blocks H = 3072 left half of them idle on the tail pass)
    for (int c0 = threadIdx.x * VEC; c0 < H; c0 += blockDim.x * VEC) {
        float acc[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] = 0.0f;
        for (long r = blockIdx.x; r < rows; r += gridDim.x) {
            float g[VEC], x[VEC], o[VEC];
            load_vec<T, VEC>(g_out + r * H + c0, g);
            load_vec<T, VEC>(h + r * H + c0, x);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                o[j] = g[j] * gelu_grad<TANH>(x[j]);
                acc[j] += to_f<T>(from_f<T>(o[j]));
            }
            store_vec<T, VEC>(g_h + r * H + c0, o);
        }
        if (partials) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) partials[(size_t)blockIdx.x * H + c0 + j] = acc[j];
        }
    }
}

// plain column sum of a [rows][H] matrix of T (qkv bias gradient): partials[block][H]
template <typename T>
__global__ __launch_bounds__(1024) void colsum_kernel(const T *__restrict__ g, long rows, int H, float *__restrict__ partials) {
    constexpr int VEC = 16 / sizeof(T);
    for (int c0 = threadIdx.x * VEC; c0 < H; c0 += blockDim.x * VEC) {
        float acc[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) acc[j] = 0.0f;
        // four rows fetched per trip before any is added (same ascending order of the adds: bit-identical sums) — one load per trip made
        // the walk a chain of dependent memory round trips (25 us for a 65664 x 2304 gradient: 12 TB/s-worth of traffic at 1.2 TB/s)
        for (long r = blockIdx.x; r < rows; r += 4L * gridDim.x) {
            float v[4][VEC];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long ru = r + (long)u * gridDim.x;
                load_vec<T, VEC>(g + (ru < rows ? ru : r) * H + c0, v[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (r + (long)u * gridDim.x < rows) {
#pragma unroll
                    for (int j = 0; j < VEC; ++j) acc[j] += v[u][j];
                }
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) partials[(size_t)blockIdx.x * H + c0 + j] = acc[j];
    }
}

// logit[r] = sum_c h[r][c] * w[c]: the 1-channel logit convolution of a discriminator head (discriminator_dino.py:215) as a row dot —
// one wave per row, 16-byte lane loads (the library route: an fp32 copy of h + a gemv; its backward: two broadcast products + a cast)
template <typename T>
__global__ __launch_bounds__(256) void rowdot_fwd_kernel(const T *__restrict__ h, const float *__restrict__ w, long rows, int C,
                                                         float *__restrict__ out) {
    constexpr int VEC = 16 / sizeof(T);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (long r = (long)blockIdx.x * 4 + wave; r < rows; r += (long)gridDim.x * 4) {
        float s = 0.0f;
        for (int c0 = lane * VEC; c0 < C; c0 += 64 * VEC) {
            float v[VEC];
            load_vec<T, VEC>(h + r * C + c0, v);
#pragma unroll
            for (int j = 0; j < VEC; ++j) s = __builtin_fmaf(v[j], w[c0 + j], s);
        }
        s = wave_sum(s);
        if (lane == 0) out[r] = s;
    }
}

// g_h[r][c] = g[r] * w[c] (T);  partials[block][c] = sum over the block's rows of g[r] * h[r][c]  (-> g_w by colsum_finalize_kernel)
template <typename T>
__global__ __launch_bounds__(1024) void rowdot_bwd_kernel(const T *__restrict__ h, const float *__restrict__ w, const float *__restrict__ g,
                                                          long rows, int C, T *__restrict__ g_h, float *__restrict__ partials) {
    constexpr int VEC = 16 / sizeof(T);
    for (int c0 = threadIdx.x * VEC; c0 < C; c0 += blockDim.x * VEC) {
        float acc[VEC], ww[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) { acc[j] = 0.0f; ww[j] = w[c0 + j]; }
        for (long r = blockIdx.x; r < rows; r += gridDim.x) {
            const float gr = g[r];
            if (partials) {
                float v[VEC];
                load_vec<T, VEC>(h + r * C + c0, v);
#pragma unroll
                for (int j = 0; j < VEC; ++j) acc[j] = __builtin_fmaf(gr, v[j], acc[j]);
            }
            if (g_h) {
                float o[VEC];
#pragma unroll
                for (int j = 0; j < VEC; ++j) o[j] = gr * ww[j];
                store_vec<T, VEC>(g_h + r * C + c0, o);
            }
        }
        if (partials) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) partials[(size_t)blockIdx.x * C + c0 + j] = acc[j];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
static int row_blocks(long rows) {
    long b = (rows + 3) / 4;
    const long cap = (long)num_cus() * 8;
    if (b > cap) b = cap;
    if (b > MAX_ROW_BLOCKS) b = MAX_ROW_BLOCKS;
    return (int)(b < 1 ? 1 : b);
}

// + FIN_CHUNKS rows behind the kernels' own: the second level of the column sums (colsum_stage_kernel)
extern "C" int xq_row_partials_blocks(int64_t rows) { return row_blocks((long)rows) + FIN_CHUNKS; }

// out[q][c] (+)= sum over the `blocks` partial rows [blocks][nq][D] a row / column kernel left in `partials` (sized by xq_row_partials_blocks)
static void launch_finalize(float *partials, int blocks, int nq, int D, float *o0, float *o1, float *o2, float *o3, int accumulate, hipStream_t s) {
    const float *src = partials;
    int n = blocks;
    if (blocks >= FIN_TWO_LEVEL_FROM && (nq * D) % 4 == 0) {      // (one level for everything: +0.3 .. +1.0 ms per train step, profiles/r04_step_ab_readout_finalize.txt)
        float *tmp = partials + (size_t)blocks * nq * D;
        hipLaunchKernelGGL(colsum_stage_kernel, dim3((nq * D / 4 + 15) / 16, FIN_CHUNKS), dim3(256), 0, s, partials, blocks, nq * D, tmp);
        src = tmp;
        n = FIN_CHUNKS;
    }
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3((nq * D + FIN_COLS - 1) / FIN_COLS), dim3(256), 0, s, src, n, nq, D, o0, o1, o2, o3, accumulate);
}

// kernels without column partials (the forward passes): the block count is only a question of bytes in flight
static int row_blocks_fwd(long rows) {
    long b = (rows + 3) / 4;
    const long cap = (long)num_cus() * XQ_FWD_BLOCKS_PER_CU;
    if (b > cap) b = cap;
    return (int)(b < 1 ? 1 : b);
}

#define DISPATCH_D(D, F)                                                           \
    switch (D) {                                                                   \
        case 64: F(1, 1); break;                                                   \
        case 128: F(1, 2); break;                                                  \
        case 256: F(1, 4); break;                                                  \
        case 384: F(3, 2); break;                                                  \
        case 512: F(2, 4); break;                                                  \
        case 768: F(3, 4); break;                                                  \
        case 1024: F(4, 4); break;                                                 \
        default: return xq_set_error(XQ_EINVAL, "%s: unsupported width D=%ld (64,128,256,384,512,768,1024)", fn, D); \
    }

extern "C" int xq_res_ln_forward(const float *x, const void *y, const float *gamma, const float *mask, int64_t rows, int D,
                                 int rows_per_sample, const float *lnw, const float *lnb, float eps, int act_bf16, float *x_new,
                                 void *a, float *mean, float *rstd, xq_stream_t stream) {
    const char *fn = "xq_res_ln_forward";
    if (rows == 0) return XQ_OK;
    if (!x || !lnw || !lnb || !a || !mean || !rstd) return xq_set_error(XQ_EINVAL, "%s: null pointer", fn);
    if (rows < 0 || rows_per_sample < 1) return xq_set_error(XQ_EINVAL, "%s: bad rows", fn);
    hipStream_t s = (hipStream_t)stream;
    int blocks = row_blocks_fwd(rows);
    static const int fwd_bpc = [] { const char *e = getenv("XQ_RES_LN_FWD_BLOCKS_PER_CU"); return e ? atoi(e) : 0; }();      // 0: the resident count
    const int pslot = xq::prof_begin(XQ_PROF_RES_LN_FWD, (double)rows * D * (4.0 + (y ? (act_bf16 ? 2.0 : 4.0) : 0.0) + (x_new ? 4.0 : 0.0) + (act_bf16 ? 2.0 : 4.0)), s);
    // grid = exactly the blocks the chip holds at once (76 VGPRs at D = 768: six blocks of four waves per CU, not the eight of XQ_FWD_BLOCKS_PER_CU —
    // a partial second round of blocks cost the backward kernel 20 %, profiles/r06_res_ln_bwd_cols.txt)
#define FWD_LAUNCH(T, NV, VEC) do { \
        auto kfn = res_ln_fwd_kernel<T, NV, VEC>; \
        static int occ = 0; \
        if (!occ) { if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kfn, ROW_THREADS, 0) != hipSuccess || occ < 1) occ = XQ_FWD_BLOCKS_PER_CU; } \
        const long cap = (long)num_cus() * (fwd_bpc > 0 ? fwd_bpc : occ); \
        if (blocks > cap) blocks = (int)cap; \
        hipLaunchKernelGGL(kfn, dim3(blocks), dim3(ROW_THREADS), 0, s, x, (const T *)y, gamma, mask, (long)rows, rows_per_sample, lnw, lnb, eps, x_new, \
                           (T *)a, mean, rstd); } while (0)
#define FWD_BF16(NV, VEC) FWD_LAUNCH(bf16, NV, VEC)
#define FWD_F32(NV, VEC) FWD_LAUNCH(float, NV, VEC)
    if (act_bf16) { DISPATCH_D(D, FWD_BF16) } else { DISPATCH_D(D, FWD_F32) }
#undef FWD_BF16
#undef FWD_F32
#undef FWD_LAUNCH
    xq::prof_end(pslot, s);
    return xq_check_launch(fn);
}

extern "C" int xq_res_ln_backward(const void *g_a, const float *g_xnew, const float *x_new, const float *mean, const float *rstd,
                                  const float *lnw, const void *y, const float *gamma, const float *mask, int64_t rows, int D,
                                  int rows_per_sample, int act_bf16, float *g_x, void *g_y, float *g_lnw, float *g_lnb,
                                  float *g_gamma, float *g_ybias, int accumulate, float *partials, xq_stream_t stream) {
    const char *fn = "xq_res_ln_backward";
    if (rows == 0) return XQ_OK;
    if (!x_new || !mean || !rstd || !lnw || !g_x || !partials) return xq_set_error(XQ_EINVAL, "%s: null pointer", fn);
    if (y && !g_y) return xq_set_error(XQ_EINVAL, "%s: g_y required when y is given", fn);
    hipStream_t s = (hipStream_t)stream;
    static const int impl = [] { const char *e = getenv("XQ_RES_LN_BWD"); return e ? atoi(e) : 1; }();        // >= 1: columns split over the waves (round 6); 0: one wave per row
    static const int bpc_env = [] { const char *e = getenv("XQ_RES_LN_BWD_BLOCKS_PER_CU"); return e ? atoi(e) : 0; }();
    int blocks = row_blocks(rows);
    const double asz = act_bf16 ? 2.0 : 4.0;
    const int pslot = xq::prof_begin(XQ_PROF_RES_LN_BWD, (double)rows * D * ((g_a ? asz : 0.0) + (g_xnew ? 4.0 : 0.0) + 4.0 + (y ? 2.0 * asz : 0.0) + 4.0), s);
    // column-split kernel: exactly the resident number of blocks (a partial second round of blocks costs 20 %: profiles/r06_res_ln_bwd_cols.txt)
#define COLS_LAUNCH(T, NV, VEC, RB, MINW) do { \
        auto kfn = res_ln_bwd_cols_kernel<T, NV, VEC, RB, MINW>; \
        static int occ = 0; \
        if (!occ) { if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kfn, NV * 64, 0) != hipSuccess || occ < 1) occ = 1; } \
        long b = (rows + RB - 1) / RB; \
        const long cap = (long)num_cus() * (bpc_env > 0 ? bpc_env : occ); \
        if (b > cap) b = cap; \
        if (b < blocks) blocks = (int)b;        /* (never more partial rows than xq_row_partials_blocks promised) */ \
        hipLaunchKernelGGL(kfn, dim3(blocks), dim3(NV * 64), 0, s, (const T *)g_a, g_xnew, x_new, mean, rstd, lnw, (const T *)y, gamma, mask, (long)rows, \
                           rows_per_sample, g_x, (T *)g_y, partials); } while (0)
    // R = 2 rows per iteration at >= 4 waves per SIMD (<= 128 VGPRs; 64 used): 65 664 x 768 0.200 -> 0.170 ms, x 384 0.122 -> 0.093 ms with the finalize;
    // R = 4: 0.176 / 0.103 (146 VGPRs, 3 waves), R = 4 forced to 128 VGPRs or R = 8: slower than the one-wave-per-row kernel
#define BWD_BF16(NV, VEC) do { if (impl >= 1) COLS_LAUNCH(bf16, NV, VEC, 2, 4); \
    else hipLaunchKernelGGL((res_ln_bwd_kernel<bf16, NV, VEC>), dim3(blocks), dim3(ROW_THREADS), 0, s, (const bf16 *)g_a, g_xnew, \
        x_new, mean, rstd, lnw, (const bf16 *)y, gamma, mask, (long)rows, rows_per_sample, g_x, (bf16 *)g_y, partials); } while (0)
#define BWD_F32(NV, VEC) do { if (impl >= 1) COLS_LAUNCH(float, NV, VEC, 2, 1); \
    else hipLaunchKernelGGL((res_ln_bwd_kernel<float, NV, VEC>), dim3(blocks), dim3(ROW_THREADS), 0, s, (const float *)g_a, g_xnew, \
        x_new, mean, rstd, lnw, (const float *)y, gamma, mask, (long)rows, rows_per_sample, g_x, (float *)g_y, partials); } while (0)
    if (act_bf16) { DISPATCH_D(D, BWD_BF16) } else { DISPATCH_D(D, BWD_F32) }
#undef BWD_BF16
#undef BWD_F32
#undef COLS_LAUNCH
    launch_finalize(partials, blocks, 4, D, g_lnw, g_lnb, g_gamma, g_ybias, accumulate, s);
    xq::prof_end(pslot, s);
    return xq_check_launch(fn);
}

extern "C" int xq_gelu_forward(const void *h, int64_t n, int act_bf16, int approximate_tanh, void *out, xq_stream_t stream) {
    const char *fn = "xq_gelu_forward";
    if (n == 0) return XQ_OK;
    if (!h || !out) return xq_set_error(XQ_EINVAL, "%s: null pointer", fn);
    const int vec = act_bf16 ? 8 : 4;
    if (n % vec) return xq_set_error(XQ_EINVAL, "%s: n=%ld must be a multiple of the 16-byte vector", fn, (long)n);
    long blocks = (n / vec + 255) / 256;
    const long cap = (long)num_cus() * 16;
    if (blocks > cap) blocks = cap;
    hipStream_t s = (hipStream_t)stream;
#define GELU_FWD(T, TANH) hipLaunchKernelGGL((gelu_fwd_kernel<T, TANH>), dim3((unsigned)blocks), dim3(256), 0, s, (const T *)h, (long)n, (T *)out)
    if (act_bf16) { if (approximate_tanh) GELU_FWD(bf16, true); else GELU_FWD(bf16, false); }
    else { if (approximate_tanh) GELU_FWD(float, true); else GELU_FWD(float, false); }
#undef GELU_FWD
    return xq_check_launch(fn);
}

extern "C" int xq_gelu_backward(const void *g_out, const void *h, int64_t rows, int H, int act_bf16, int approximate_tanh, void *g_h,
                                float *g_bias, int accumulate, float *partials, xq_stream_t stream) {
    const char *fn = "xq_gelu_backward";
    if (rows == 0) return XQ_OK;
    if (!g_out || !h || !g_h) return xq_set_error(XQ_EINVAL, "%s: null pointer", fn);
    const int vec = act_bf16 ? 8 : 4;
    if (H % vec) return xq_set_error(XQ_EINVAL, "%s: H=%ld must be a multiple of the 16-byte vector", fn, H);
    if (g_bias && !partials) return xq_set_error(XQ_EINVAL, "%s: partials workspace required for the bias gradient", fn);
    const int blocks = row_blocks(rows * 4);
    hipStream_t s = (hipStream_t)stream;
    const int threads = col_threads(H / vec);
#define GELU_BWD(T, TANH) hipLaunchKernelGGL((gelu_bwd_kernel<T, TANH>), dim3(blocks), dim3(threads), 0, s, (const T *)g_out, (const T *)h, (long)rows, H, \
                                             (T *)g_h, g_bias ? partials : nullptr)
    if (act_bf16) { if (approximate_tanh) GELU_BWD(bf16, true); else GELU_BWD(bf16, false); }
    else { if (approximate_tanh) GELU_BWD(float, true); else GELU_BWD(float, false); }
#undef GELU_BWD
    if (g_bias)
        launch_finalize(partials, blocks, 1, H, g_bias, nullptr, nullptr, nullptr, accumulate, s);
    return xq_check_launch(fn);
}

extern "C" int xq_colsum(const void *g, int64_t rows, int H, int act_bf16, float *out, int accumulate, float *partials,
                         xq_stream_t stream) {
    const char *fn = "xq_colsum";
    if (rows == 0) return XQ_OK;
    if (!g || !out || !partials) return xq_set_error(XQ_EINVAL, "%s: null pointer", fn);
    const int vec = act_bf16 ? 8 : 4;
    if (H % vec) return xq_set_error(XQ_EINVAL, "%s: H=%ld must be a multiple of the 16-byte vector", fn, H);
    const int blocks = row_blocks(rows * 4);
    hipStream_t s = (hipStream_t)stream;
    const int threads = col_threads(H / vec);
    if (act_bf16) hipLaunchKernelGGL((colsum_kernel<bf16>), dim3(blocks), dim3(threads), 0, s, (const bf16 *)g, (long)rows, H, partials);
    else hipLaunchKernelGGL((colsum_kernel<float>), dim3(blocks), dim3(threads), 0, s, (const float *)g, (long)rows, H, partials);
    launch_finalize(partials, blocks, 1, H, out, nullptr, nullptr, nullptr, accumulate, s);
    return xq_check_launch(fn);
}

// ================================================================================================
// LPIPS feature comparison (reference lpips.py:85-96,159-164), one fused pass per VGG level:
//   val[b] = (1/HW) sum_{h,w} sum_c w_c * ( f0/(|f0|+eps) - f1/(|f1|+eps) )^2 ,  |f| = sqrt(sum_c f^2), eps = 1e-10
// f0, f1: channels-last activations, i.e. [B][HW][C] contiguous (T = bf16 under autocast, fp32 otherwise); w: [C] fp32.
// The reference runs ~10 elementwise/reduction kernels over fp32 copies of both feature maps per level (1 G elements
// per image set at B = 128); here each level reads f0 and f1 once (forward) and once more + writes d f1 (backward).
// A pixel is handled by C/8 consecutive lanes (8 channels = one 16-byte load per lane for bf16).
// ================================================================================================
template <typename T, int C>
__global__ __launch_bounds__(256) void lpips_level_fwd_kernel(const T *__restrict__ f0, const T *__restrict__ f1,
                                                              const float *__restrict__ w, int HW, float *__restrict__ val) {
    constexpr int LPP = C / 8;  // lanes per pixel (8, 16, 32, 64)
    constexpr int PPW = 64 / LPP;  // pixels per wave
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane % LPP, pix_in_wave = lane / LPP;
    float wv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) wv[j] = w[sub * 8 + j];
    float acc = 0.0f;
    const size_t base = (size_t)b * HW * C;
    for (int p = (blockIdx.x * 4 + wave) * PPW + pix_in_wave; p < HW; p += gridDim.x * 4 * PPW) {
        float a[8], c[8];
        load_vec<T, 8>(f0 + base + (size_t)p * C + sub * 8, a);
        load_vec<T, 8>(f1 + base + (size_t)p * C + sub * 8, c);
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { s0 = __builtin_fmaf(a[j], a[j], s0); s1 = __builtin_fmaf(c[j], c[j], s1); }
#pragma unroll
        for (int o = LPP / 2; o > 0; o >>= 1) { s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); }
        const float i0 = 1.0f / (__builtin_sqrtf(s0) + 1e-10f), i1 = 1.0f / (__builtin_sqrtf(s1) + 1e-10f);
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = a[j] * i0 - c[j] * i1; acc = __builtin_fmaf(wv[j] * d, d, acc); }
    }
    __shared__ float red[4];
    acc = wave_sum(acc);
    if (lane == 0) red[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(val + b, ((red[0] + red[1]) + (red[2] + red[3])) / (float)HW);
}

// g1 = d val[b]/d f1 * gout[b]:  g_u = -2 w (u0 - u1) * gout/HW ;  g_f = g_u/n - f * (g_u . f) / (n^2 r),  n = r + eps
template <typename T, int C>
__global__ __launch_bounds__(256) void lpips_level_bwd_kernel(const T *__restrict__ f0, const T *__restrict__ f1,
                                                              const float *__restrict__ w, const float *__restrict__ gout,
                                                              const T *__restrict__ g_add, int relu_mask, int HW, T *__restrict__ g1) {
    constexpr int LPP = C / 8;
    constexpr int PPW = 64 / LPP;
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane % LPP, pix_in_wave = lane / LPP;
    float wv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) wv[j] = w[sub * 8 + j];
    const float gs = gout[b] / (float)HW;
    const size_t base = (size_t)b * HW * C;
    for (int p = (blockIdx.x * 4 + wave) * PPW + pix_in_wave; p < HW; p += gridDim.x * 4 * PPW) {
        float a[8], c[8];
        load_vec<T, 8>(f0 + base + (size_t)p * C + sub * 8, a);
        load_vec<T, 8>(f1 + base + (size_t)p * C + sub * 8, c);
        float s0 = 0.f, s1 = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { s0 = __builtin_fmaf(a[j], a[j], s0); s1 = __builtin_fmaf(c[j], c[j], s1); }
#pragma unroll
        for (int o = LPP / 2; o > 0; o >>= 1) { s0 += __shfl_xor(s0, o); s1 += __shfl_xor(s1, o); }
        const float r1 = __builtin_sqrtf(s1);
        const float i0 = 1.0f / (__builtin_sqrtf(s0) + 1e-10f), n1 = r1 + 1e-10f, i1 = 1.0f / n1;
        float gu[8], dot = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            gu[j] = -2.0f * wv[j] * (a[j] * i0 - c[j] * i1) * gs;
            dot = __builtin_fmaf(gu[j], c[j], dot);
        }
#pragma unroll
        for (int o = LPP / 2; o > 0; o >>= 1) dot += __shfl_xor(dot, o);
        const float k = r1 > 0.0f ? dot * i1 * i1 / r1 : 0.0f;
        float out[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) out[j] = gu[j] * i1 - c[j] * k;
        if (g_add) {       // gradient from the deeper VGG slices (hand-driven backward: xq_lpips_level_backward_fused)
            float ga[8];
            load_vec<T, 8>(g_add + base + (size_t)p * C + sub * 8, ga);
#pragma unroll
            for (int j = 0; j < 8; ++j) out[j] += ga[j];
        }
        if (relu_mask) {   // f1 is a ReLU output: gradient w.r.t. the pre-activation
#pragma unroll
            for (int j = 0; j < 8; ++j) out[j] = c[j] > 0.0f ? out[j] : 0.0f;
        }
        store_vec<T, 8>(g1 + base + (size_t)p * C + sub * 8, out);
    }
}

template <typename T>
static int lpips_dispatch(bool bwd, const T *f0, const T *f1, const float *w, const float *gout, int B, int HW, int C, float *val,
                          T *g1, hipStream_t s, const T *g_add = nullptr, int relu_mask = 0) {
    int bx = (HW + 31) / 32;
    const int cap = (num_cus() * 8 + B - 1) / B;
    if (bx > cap) bx = cap < 1 ? 1 : cap;
    dim3 grid(bx, B), block(256);
#define LP_CASE(CC)                                                                                                      \
    case CC:                                                                                                             \
        if (bwd) hipLaunchKernelGGL((lpips_level_bwd_kernel<T, CC>), grid, block, 0, s, f0, f1, w, gout, g_add, relu_mask, HW, g1);        \
        else hipLaunchKernelGGL((lpips_level_fwd_kernel<T, CC>), grid, block, 0, s, f0, f1, w, HW, val);                 \
        break;
    switch (C) {
        LP_CASE(64) LP_CASE(128) LP_CASE(256) LP_CASE(512)
        default: return xq_set_error(XQ_EINVAL, "%s: unsupported channel count C=%ld (64,128,256,512)", "xq_lpips_level", C);
    }
#undef LP_CASE
    return xq_check_launch("lpips_level kernel");
}

extern "C" int xq_lpips_level_forward(const void *f0, const void *f1, const float *w, int B, int HW, int C, int act_bf16, float *val,
                                      xq_stream_t stream) {
    if (B == 0) return XQ_OK;
    if (!f0 || !f1 || !w || !val) return xq_set_error(XQ_EINVAL, "%s: null pointer", "xq_lpips_level_forward");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(val, 0, (size_t)B * 4, s) != hipSuccess) return xq_set_error(XQ_ELAUNCH, "%s", "hipMemsetAsync failed");
    if (act_bf16) return lpips_dispatch<bf16>(false, (const bf16 *)f0, (const bf16 *)f1, w, nullptr, B, HW, C, val, nullptr, s);
    return lpips_dispatch<float>(false, (const float *)f0, (const float *)f1, w, nullptr, B, HW, C, val, nullptr, s);
}

extern "C" int xq_lpips_level_backward(const void *f0, const void *f1, const float *w, const float *gout, int B, int HW, int C,
                                       int act_bf16, void *g1, xq_stream_t stream) {
    if (B == 0) return XQ_OK;
    if (!f0 || !f1 || !w || !gout || !g1) return xq_set_error(XQ_EINVAL, "%s: null pointer", "xq_lpips_level_backward");
    hipStream_t s = (hipStream_t)stream;
    if (act_bf16) return lpips_dispatch<bf16>(true, (const bf16 *)f0, (const bf16 *)f1, w, gout, B, HW, C, nullptr, (bf16 *)g1, s);
    return lpips_dispatch<float>(true, (const float *)f0, (const float *)f1, w, gout, B, HW, C, nullptr, (float *)g1, s);
}

extern "C" int xq_lpips_level_backward_fused(const void *f0, const void *f1, const float *w, const float *gout, const void *g_add, int relu_mask,
                                             int B, int HW, int C, int act_bf16, void *g1, xq_stream_t stream) {
    if (B == 0) return XQ_OK;
    if (!f0 || !f1 || !w || !gout || !g1) return xq_set_error(XQ_EINVAL, "%s: null pointer", "xq_lpips_level_backward_fused");
    hipStream_t s = (hipStream_t)stream;
    if (act_bf16)
        return lpips_dispatch<bf16>(true, (const bf16 *)f0, (const bf16 *)f1, w, gout, B, HW, C, nullptr, (bf16 *)g1, s, (const bf16 *)g_add, relu_mask);
    return lpips_dispatch<float>(true, (const float *)f0, (const float *)f1, w, gout, B, HW, C, nullptr, (float *)g1, s, (const float *)g_add, relu_mask);
}

extern "C" int xq_rowdot_forward(const void *h, const float *w, int64_t rows, int C, int act_bf16, float *out, xq_stream_t stream) {
    const char *fn = "xq_rowdot_forward";
    if (rows == 0) return XQ_OK;
    if (!h || !w || !out) return xq_set_error(XQ_EINVAL, "%s: null pointer", fn);
    const int vec = act_bf16 ? 8 : 4;
    if (C % vec) return xq_set_error(XQ_EINVAL, "%s: C=%ld must be a multiple of the 16-byte vector", fn, (long)C);
    const int blocks = row_blocks_fwd(rows);
    hipStream_t s = (hipStream_t)stream;
    if (act_bf16) hipLaunchKernelGGL((rowdot_fwd_kernel<bf16>), dim3(blocks), dim3(256), 0, s, (const bf16 *)h, w, (long)rows, C, out);
    else hipLaunchKernelGGL((rowdot_fwd_kernel<float>), dim3(blocks), dim3(256), 0, s, (const float *)h, w, (long)rows, C, out);
    return xq_check_launch(fn);
}

/* g_h (nullable) [rows][C] = g[r] w[c]; g_w (nullable) [C] = sum_r g[r] h[r][c] (partials: xq_row_partials_blocks(rows * 4) x C floats) */
extern "C" int xq_rowdot_backward(const void *h, const float *w, const float *g, int64_t rows, int C, int act_bf16, void *g_h, float *g_w,
                                  float *partials, xq_stream_t stream) {
    const char *fn = "xq_rowdot_backward";
    if (rows == 0) return XQ_OK;
    if (!h || !w || !g) return xq_set_error(XQ_EINVAL, "%s: null pointer", fn);
    if (g_w && !partials) return xq_set_error(XQ_EINVAL, "%s: partials workspace required for g_w", fn);
    const int vec = act_bf16 ? 8 : 4;
    if (C % vec) return xq_set_error(XQ_EINVAL, "%s: C=%ld must be a multiple of the 16-byte vector", fn, (long)C);
    const int blocks = row_blocks(rows * 4);
    hipStream_t s = (hipStream_t)stream;
    const int threads = col_threads(C / vec);
    float *part = g_w ? partials : nullptr;
    if (act_bf16) hipLaunchKernelGGL((rowdot_bwd_kernel<bf16>), dim3(blocks), dim3(threads), 0, s, (const bf16 *)h, w, g, (long)rows, C, (bf16 *)g_h, part);
    else hipLaunchKernelGGL((rowdot_bwd_kernel<float>), dim3(blocks), dim3(threads), 0, s, (const float *)h, w, g, (long)rows, C, (float *)g_h, part);
    if (g_w)
        launch_finalize(partials, blocks, 1, C, g_w, nullptr, nullptr, nullptr, 0, s);
    return xq_check_launch(fn);
}

/* out[c] = sum over r < nrows of partials[r][c] in a fixed order (the per-tile column partials the fused-GELU data-gradient GEMM leaves for
 * the fc1 bias gradient: xq_gemm_bf16_nn_gelu_bwd) */
extern "C" int xq_colsum_partials(const float *partials, int nrows, int D, float *out, xq_stream_t stream) {
    if (D <= 0) return XQ_OK;
    if (!partials || !out || nrows < 0) return xq_set_error(XQ_EINVAL, "%s: bad arguments", "xq_colsum_partials");
    hipLaunchKernelGGL(colsum_finalize_kernel, dim3((D + FIN_COLS - 1) / FIN_COLS), dim3(256), 0, (hipStream_t)stream, partials, nrows, 1, D, out,
                       nullptr, nullptr, nullptr, 0);
    return xq_check_launch("xq_colsum_partials");
}

// ================================================================================================
// Round 5: token assembly in front of a block stack (dino_enc/dinov2.py:149-179 encoder, :313-349 decoder; vision_transformer.py:818-851
// _pos_embed): x[b][t] = table[t] + (start <= t < start + n ? data[b][t - start] : 0), where `table` (N x D fp32) is everything that
// does not depend on the sample — class token, position table (resampled for the latent grid), learnable latent / mask tokens, level
// embedding, assembled by the module's own op chain on a batch of ONE — and `data` are the per-sample tokens (patch embeddings in the
// encoder and the teacher, quantised latents in the decoder).  Upstream then casts x to the autocast dtype; with round_bf16 the output is
// that rounding, kept in fp32 (what the first LayerNorm / residual kernels read): the op chain's cat / add / cat / add / cast / cast-back
// passes over a (B, N, D) tensor — six to eight of them per stack and direction, ~2.6 ms of a 180 ms step — become one pass.
// Backward: g_data = the slice of g (in data's dtype), g_table[t] = sum_b g[b][t] in ascending b (deterministic) — one pass as well.
// ================================================================================================
template <typename TD>
__global__ __launch_bounds__(256) void token_assemble_fwd_kernel(const float *__restrict__ table, const TD *__restrict__ data, int B, int N, int n, int start,
                                                                 int D, int round_bf16, float *__restrict__ out) {
    const int dv = D / 4;
    const long total = (long)B * N * dv;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int d4 = (int)(i % dv);
        const long bt = i / dv;
        const int t = (int)(bt % N);
        const long b = bt / N;
        float4 v = *reinterpret_cast<const float4 *>(table + (long)t * D + 4 * d4);
        if (t >= start && t < start + n) {
            float e[4];
            load_vec<TD, 4>(data + ((b * n + (t - start)) * (long)D + 4 * d4), e);
            v.x += e[0]; v.y += e[1]; v.z += e[2]; v.w += e[3];
        }
        if (round_bf16) {
            v.x = __bfloat162float(__float2bfloat16(v.x)); v.y = __bfloat162float(__float2bfloat16(v.y));
            v.z = __bfloat162float(__float2bfloat16(v.z)); v.w = __bfloat162float(__float2bfloat16(v.w));
        }
        *reinterpret_cast<float4 *>(out + i * 4) = v;
    }
}

template <typename TD>
__global__ __launch_bounds__(256) void token_assemble_bwd_kernel(const float *__restrict__ g, int B, int N, int n, int start, int D, TD *__restrict__ g_data,
                                                                 float *__restrict__ g_table) {
    const int dv = D / 4;
    const long total = (long)N * dv;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int d4 = (int)(i % dv);
        const int t = (int)(i / dv);
        const bool has = g_data != nullptr && t >= start && t < start + n;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int b0 = 0; b0 < B; b0 += 4) {      // four samples in flight per trip, added in ascending order
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int b = b0 + u < B ? b0 + u : b0;
                v[u] = *reinterpret_cast<const float4 *>(g + (((long)b * N + t) * D + 4 * d4));
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (b0 + u < B) {
                    acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w;
                    if (has) {
                        const float e[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
                        store_vec<TD, 4>(g_data + (((long)(b0 + u) * n + (t - start)) * D + 4 * d4), e);
                    }
                }
        }
        if (g_table) *reinterpret_cast<float4 *>(g_table + (long)t * D + 4 * d4) = acc;
    }
}

extern "C" int xq_token_assemble_forward(const float *table, const void *data, int data_bf16, int B, int N, int n, int start, int D, int round_bf16,
                                         float *out, xq_stream_t stream) {
    const char *fn = "xq_token_assemble_forward";
    if (B < 0 || N < 1 || n < 0 || start < 0 || start + n > N || D < 4 || D % 4) return xq_set_error(XQ_EINVAL, "%s: bad geometry (D %% 4, start + n <= N)", fn);
    if (B == 0) return XQ_OK;
    if (!table || !out || (n > 0 && !data)) return xq_set_error(XQ_EINVAL, "%s: null pointer", fn);
    const long total = (long)B * N * (D / 4);
    long blocks = (total + 255) / 256;
    const long cap = (long)num_cus() * 16;
    if (blocks > cap) blocks = cap;
    hipStream_t s = (hipStream_t)stream;
    if (data_bf16) hipLaunchKernelGGL((token_assemble_fwd_kernel<bf16>), dim3((unsigned)blocks), dim3(256), 0, s, table, (const bf16 *)data, B, N, n, start, D, round_bf16, out);
    else hipLaunchKernelGGL((token_assemble_fwd_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, s, table, (const float *)data, B, N, n, start, D, round_bf16, out);
    return xq_check_launch(fn);
}

extern "C" int xq_token_assemble_backward(const float *g, int data_bf16, int B, int N, int n, int start, int D, void *g_data, float *g_table,
                                          xq_stream_t stream) {
    const char *fn = "xq_token_assemble_backward";
    if (B < 0 || N < 1 || n < 0 || start < 0 || start + n > N || D < 4 || D % 4) return xq_set_error(XQ_EINVAL, "%s: bad geometry (D %% 4, start + n <= N)", fn);
    if (B == 0 || (!g_data && !g_table)) return XQ_OK;
    if (!g) return xq_set_error(XQ_EINVAL, "%s: null pointer", fn);
    const long total = (long)N * (D / 4);
    const long blocks = (total + 255) / 256;
    hipStream_t s = (hipStream_t)stream;
    if (data_bf16) hipLaunchKernelGGL((token_assemble_bwd_kernel<bf16>), dim3((unsigned)blocks), dim3(256), 0, s, g, B, N, n, start, D, (bf16 *)g_data, g_table);
    else hipLaunchKernelGGL((token_assemble_bwd_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, s, g, B, N, n, start, D, (float *)g_data, g_table);
    return xq_check_launch(fn);
}
