// xq_perturb.hip — RobustTok latent perturbation on gfx950.
//
// Replaces reference add_perturbation (tokenizer/tokenizer_image/latent_perturbation.py:4-35):
//   d = |zhat|^2 + |ehat|^2 - 2 zhat.ehat (:16-18); topk(d, delta, largest=False) (:20); pick rank
//   r_n = (rand_n > alpha) ? 0 : randint_n (:21-24); z' = zhat + sg(norm(E[pick]) - zhat) (:26-30);
//   out = where(sample < int(B*beta), z', z_q) (:32-35).
// The reference computes the N x V matrix and a full top-delta for ALL tokens and then keeps the first
// int(B*beta) samples; here only those samples are touched and only the needed rank is selected:
//   K_dist   : distance rows for the perturbed tokens (same fp32-MFMA chain and expression as the assign
//              kernel, A1-A3), written to a workspace in token chunks;
//   K_select : one 256-thread block per token, 4-pass 8-bit radix select on the order-preserving uint32
//              image of d for the rank-r value, then the (r - #smaller)-th lowest index among equal values
//              (ties -> lower index first, matching torch.topk(sorted=True) as observed on CPU);
//   K_finish : gather + renormalise + straight-through for perturbed samples, copy of z_q for the rest.
// Backward: perturbed samples send g_out through the l2-normalise Jacobian to h, all others to z_q.
#include "xq_common.hpp"
#include "xq_internal.hpp"
#include "../../include/xq_ops.h"

#include <stdio.h>

using namespace xq;

static constexpr int DIST_TOK_PER_BLOCK = 128;  // 4 waves x 32 tokens
static constexpr int PERT_CHUNK_TOKENS = 8192;  // distance rows materialised per pass

// d rows for tokens [t_begin, t_begin + T): D[(t - t_begin) * Vpad + code]
template <int C, int MODE>
__global__ __launch_bounds__(256) void dist_rows_kernel(const float *__restrict__ z, long N, int HW, long t_begin, long T,
                                                        const float *__restrict__ wb, const float *__restrict__ ee,
                                                        int n_tiles, int Vpad, float *__restrict__ D) {
    constexpr int KQ = C / 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, li = lane & 31;
    const long tok0 = t_begin + (long)blockIdx.x * DIST_TOK_PER_BLOCK + wave * 32;
    float a[C / 2];
    float zzr[16];
    load_tokens<C, MODE>(z, N, HW, tok0, lane, a, zzr);
    const float4 *wb4 = reinterpret_cast<const float4 *>(wb);
    for (int tile = blockIdx.y; tile < n_tiles; tile += gridDim.y) {
        const float4 *src = wb4 + (size_t)tile * (KQ * 64) + lane;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
        for (int q = 0; q < KQ; ++q) {
            const float4 bv = src[q * 64];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * q + 0], bv.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * q + 1], bv.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * q + 2], bv.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4 * q + 3], bv.w, acc, 0, 0, 0);
        }
        const float e = ee[(size_t)tile * 32 + li];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long tn = tok0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            float d;
            if (MODE == MODE_COSINE) d = e - acc[r];
            else d = __builtin_fmaf(-2.0f, acc[r], zzr[r] + e);
            if (tn < t_begin + T) D[(size_t)(tn - t_begin) * Vpad + tile * 32 + li] = d;
        }
    }
}

// rank selection: sel[t] = index of the rank[t]-th smallest of D[t][0..V) under (value, index) order
__global__ __launch_bounds__(256) void select_rank_kernel(const float *__restrict__ D, int V, int Vpad,
                                                          const int32_t *__restrict__ rank, int64_t *__restrict__ sel) {
    __shared__ unsigned hist[256];
    __shared__ unsigned sh_prefix, sh_rank;
    __shared__ unsigned wsum[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float *row = D + (size_t)blockIdx.x * Vpad;
    int r0 = rank[blockIdx.x];
    if (r0 < 0) r0 = 0;
    if (r0 > V - 1) r0 = V - 1;
    if (tid == 0) { sh_prefix = 0u; sh_rank = (unsigned)r0; }
    __syncthreads();
    // 4 passes, 8 bits each, most significant first
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        hist[tid] = 0u;
        __syncthreads();
        const unsigned prefix = sh_prefix;
        const unsigned pmask = pass == 0 ? 0u : (0xffffffffu << (shift + 8));
        for (int j = tid; j < V; j += 256) {
            const unsigned k = f2ord(row[j]);
            if ((k & pmask) == prefix) atomicAdd(&hist[(k >> shift) & 255u], 1u);
        }
        __syncthreads();
        // bucket holding the wanted rank: thread t owns bin t; inclusive scan over 256 bins
        const unsigned c = hist[tid];
        unsigned incl = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned v = __shfl_up(incl, o);
            if (lane >= o) incl += v;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        unsigned base = 0;
        for (int w = 0; w < wave; ++w) base += wsum[w];
        incl += base;
        const unsigned excl = incl - c;
        const unsigned want = sh_rank;
        __syncthreads();
        if (want >= excl && want < incl) {  // exactly one thread
            sh_prefix = prefix | ((unsigned)tid << shift);
            sh_rank = want - excl;
        }
        __syncthreads();
    }
    // sh_prefix = the rank-th key value; sh_rank = how many equal keys with a lower index precede the answer
    const unsigned key = sh_prefix;
    const unsigned skip = sh_rank;
    const int per = (V + 255) / 256;
    const int j0 = tid * per;
    int j1 = j0 + per;
    if (j1 > V) j1 = V;
    unsigned cnt = 0;
    for (int j = j0; j < j1; ++j) cnt += (f2ord(row[j]) == key) ? 1u : 0u;
    unsigned incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned v = __shfl_up(incl, o);
        if (lane >= o) incl += v;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    unsigned base = 0;
    for (int w = 0; w < wave; ++w) base += wsum[w];
    incl += base;
    const unsigned excl = incl - cnt;
    if (skip >= excl && skip < incl) {
        unsigned left = skip - excl;
        for (int j = j0; j < j1; ++j) {
            if (f2ord(row[j]) == key) {
                if (left == 0) { sel[blockIdx.x] = (int64_t)j; break; }
                --left;
            }
        }
    }
}

// out = where(sample < n_pert, zhat + (norm(E[sel]) - zhat), zq_in)
template <int C, bool NORMED>
__global__ __launch_bounds__(256) void perturb_finish_kernel(const float *__restrict__ z, const float *__restrict__ zq_in,
                                                             long N, int HW, const float *__restrict__ E,
                                                             const int64_t *__restrict__ sel, long n_pert_tokens,
                                                             float *__restrict__ out) {
    const long n = (long)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const long b = n / HW;
    const int hw = (int)(n - b * HW);
    const size_t off = (size_t)b * C * HW + hw;
    if (n >= n_pert_tokens) {
#pragma unroll
        for (int k = 0; k < C; ++k) out[off + (size_t)k * HW] = zq_in[off + (size_t)k * HW];
        return;
    }
    float x[C], zh[C], e[C], eh[C];
#pragma unroll
    for (int k = 0; k < C; ++k) x[k] = z[off + (size_t)k * HW];
    const float4 *row = reinterpret_cast<const float4 *>(E + (size_t)sel[n] * C);
#pragma unroll
    for (int q = 0; q < C / 4; ++q) {
        float4 v = row[q];
        e[4 * q + 0] = v.x; e[4 * q + 1] = v.y; e[4 * q + 2] = v.z; e[4 * q + 3] = v.w;
    }
    if (NORMED) {
        l2norm_row<C>(x, zh);
        l2norm_row<C>(e, eh);
    } else {
#pragma unroll
        for (int k = 0; k < C; ++k) { zh[k] = x[k]; eh[k] = e[k]; }
    }
#pragma unroll
    for (int k = 0; k < C; ++k) out[off + (size_t)k * HW] = zh[k] + (eh[k] - zh[k]);
}

template <int C, bool NORMED>
__global__ __launch_bounds__(256) void perturb_backward_kernel(const float *__restrict__ z, long N, int HW,
                                                               long n_pert_tokens, const float *__restrict__ g_out,
                                                               float *__restrict__ g_z, float *__restrict__ g_zq) {
    const long n = (long)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const long b = n / HW;
    const int hw = (int)(n - b * HW);
    const size_t off = (size_t)b * C * HW + hw;
    if (n >= n_pert_tokens) {
#pragma unroll
        for (int k = 0; k < C; ++k) {
            g_zq[off + (size_t)k * HW] = g_out[off + (size_t)k * HW];
            g_z[off + (size_t)k * HW] = 0.0f;
        }
        return;
    }
    float x[C], zh[C], g[C];
#pragma unroll
    for (int k = 0; k < C; ++k) { x[k] = z[off + (size_t)k * HW]; g[k] = g_out[off + (size_t)k * HW]; }
    float nz = 1.0f;
    if (NORMED) nz = l2norm_row<C>(x, zh);
    float dz = 0.0f;
#pragma unroll
    for (int k = 0; k < C; ++k) dz = __builtin_fmaf(g[k], NORMED ? zh[k] : 0.0f, dz);
#pragma unroll
    for (int k = 0; k < C; ++k) {
        g_z[off + (size_t)k * HW] = NORMED ? (g[k] - zh[k] * dz) / nz : g[k];
        g_zq[off + (size_t)k * HW] = 0.0f;
    }
}

// ------------------------------------------------------------------------------------------------
struct PerturbWs {
    AssignWs aw;  // codebook part (keys unused)
    float *D;
    int64_t *sel;
    long chunk_tokens;
};

static size_t perturb_ws_layout(long T, int C, int V, char *base, PerturbWs *ws) {
    AssignWs aw;
    size_t off = assign_ws_layout(0, C, V, base, &aw);
    const long chunk = T < PERT_CHUNK_TOKENS ? T : PERT_CHUNK_TOKENS;
    const size_t o_d = off;
    off = align_up(off + (size_t)chunk * aw.Vpad * 4, 256);
    const size_t o_sel = off;
    off = align_up(off + (size_t)T * 8, 256);
    if (ws) {
        ws->aw = aw;
        ws->D = (float *)(base + o_d);
        ws->sel = (int64_t *)(base + o_sel);
        ws->chunk_tokens = chunk;
    }
    return off;
}

extern "C" size_t xq_perturb_workspace_bytes(int64_t n_pert_tokens, int C, int V) {
    if (n_pert_tokens < 0 || C < 1 || V < 1) return 0;
    return perturb_ws_layout((long)n_pert_tokens, C, V, nullptr, nullptr);
}

template <int C>
static int launch_dist(int mode, const float *z, long N, int HW, long t0, long T, const PerturbWs &ws, hipStream_t s) {
    const int n_tiles = ws.aw.Vpad / 32;
    const int bx = (int)((T + DIST_TOK_PER_BLOCK - 1) / DIST_TOK_PER_BLOCK);
    int by = (2 * num_cus() + bx - 1) / bx;
    if (by < 1) by = 1;
    if (by > n_tiles) by = n_tiles;
    dim3 grid(bx, by), block(256);
    switch (mode) {
        case XQ_MODE_L2_NORMED:
            hipLaunchKernelGGL((dist_rows_kernel<C, MODE_L2_NORMED>), grid, block, 0, s, z, N, HW, t0, T, ws.aw.wb, ws.aw.ee, n_tiles, ws.aw.Vpad, ws.D);
            break;
        case XQ_MODE_L2_RAW:
            hipLaunchKernelGGL((dist_rows_kernel<C, MODE_L2_RAW>), grid, block, 0, s, z, N, HW, t0, T, ws.aw.wb, ws.aw.ee, n_tiles, ws.aw.Vpad, ws.D);
            break;
        default:
            hipLaunchKernelGGL((dist_rows_kernel<C, MODE_COSINE>), grid, block, 0, s, z, N, HW, t0, T, ws.aw.wb, ws.aw.ee, n_tiles, ws.aw.Vpad, ws.D);
    }
    return xq_check_launch("dist_rows_kernel");
}

template <int C>
static int perturb_forward_c(const float *z, const float *zq_in, const float *E, long N, int HW, int V, int normed, long Tp,
                             const int32_t *rank, float *out, int64_t *sel_out, const PerturbWs &ws, hipStream_t s) {
    const int mode = normed ? XQ_MODE_L2_NORMED : XQ_MODE_L2_RAW;
    if (Tp > 0) {
        int rc = launch_assign(mode, C, z, N, HW, E, V, ws.aw, s, XQI_PREP);
        if (rc) return rc;
        for (long t0 = 0; t0 < Tp; t0 += ws.chunk_tokens) {
            const long T = (Tp - t0 < ws.chunk_tokens) ? (Tp - t0) : ws.chunk_tokens;
            rc = launch_dist<C>(mode, z, N, HW, t0, T, ws, s);
            if (rc) return rc;
            hipLaunchKernelGGL(select_rank_kernel, dim3((unsigned)T), dim3(256), 0, s, ws.D, V, ws.aw.Vpad, rank + t0, ws.sel + t0);
            rc = xq_check_launch("select_rank_kernel");
            if (rc) return rc;
        }
        if (sel_out) {
            if (hipMemcpyAsync(sel_out, ws.sel, (size_t)Tp * 8, hipMemcpyDeviceToDevice, s) != hipSuccess)
                return xq_set_error(XQ_ELAUNCH, "%s", "hipMemcpyAsync(sel) failed");
        }
    }
    const unsigned blocks = (unsigned)((N + 255) / 256);
    if (normed)
        hipLaunchKernelGGL((perturb_finish_kernel<C, true>), dim3(blocks), dim3(256), 0, s, z, zq_in, N, HW, E, ws.sel, Tp, out);
    else
        hipLaunchKernelGGL((perturb_finish_kernel<C, false>), dim3(blocks), dim3(256), 0, s, z, zq_in, N, HW, E, ws.sel, Tp, out);
    return xq_check_launch("perturb_finish_kernel");
}

extern "C" int xq_perturb_forward(const float *z, const float *zq_in, const float *E, int B, int C, int HW, int V,
                                  int codebook_norm, int n_pert, const int32_t *rank, float *out, int64_t *sel_idx,
                                  void *workspace, size_t workspace_bytes, xq_stream_t stream) {
    int rc = check_common("xq_perturb_forward", z, B, C, HW, E, V);
    if (rc) return rc;
    if (B == 0) return XQ_OK;
    if (!zq_in || !out) return xq_set_error(XQ_EINVAL, "%s: null zq_in/out", "xq_perturb_forward");
    if (n_pert < 0 || n_pert > B) return xq_set_error(XQ_EINVAL, "%s: n_pert=%ld out of range", "xq_perturb_forward", n_pert);
    if (n_pert > 0 && !rank) return xq_set_error(XQ_EINVAL, "%s: rank is null", "xq_perturb_forward");
    const long N = (long)B * HW, Tp = (long)n_pert * HW;
    PerturbWs ws;
    const size_t need = perturb_ws_layout(Tp, C, V, (char *)workspace, &ws);
    if (!workspace || workspace_bytes < need)
        return xq_set_error(XQ_ENOSPACE, "%s: workspace %ld < %ld bytes", "xq_perturb_forward", (long)workspace_bytes, (long)need);
    hipStream_t s = (hipStream_t)stream;
    switch (C) {
        case 8: return perturb_forward_c<8>(z, zq_in, E, N, HW, V, codebook_norm, Tp, rank, out, sel_idx, ws, s);
        case 16: return perturb_forward_c<16>(z, zq_in, E, N, HW, V, codebook_norm, Tp, rank, out, sel_idx, ws, s);
        case 32: return perturb_forward_c<32>(z, zq_in, E, N, HW, V, codebook_norm, Tp, rank, out, sel_idx, ws, s);
        case 64: return perturb_forward_c<64>(z, zq_in, E, N, HW, V, codebook_norm, Tp, rank, out, sel_idx, ws, s);
    }
    return XQ_EINVAL;
}

template <int C>
static void launch_pbwd(bool normed, const float *z, long N, int HW, long Tp, const float *g_out, float *g_z, float *g_zq, hipStream_t s) {
    const unsigned blocks = (unsigned)((N + 255) / 256);
    if (normed) hipLaunchKernelGGL((perturb_backward_kernel<C, true>), dim3(blocks), dim3(256), 0, s, z, N, HW, Tp, g_out, g_z, g_zq);
    else hipLaunchKernelGGL((perturb_backward_kernel<C, false>), dim3(blocks), dim3(256), 0, s, z, N, HW, Tp, g_out, g_z, g_zq);
}

extern "C" int xq_perturb_backward(const float *z, int B, int C, int HW, int codebook_norm, int n_pert, const float *g_out,
                                   float *g_z, float *g_zq, xq_stream_t stream) {
    if (B == 0) return XQ_OK;
    if (!z || !g_out || !g_z || !g_zq) return xq_set_error(XQ_EINVAL, "%s: null pointer", "xq_perturb_backward");
    if (C != 8 && C != 16 && C != 32 && C != 64) return xq_set_error(XQ_EINVAL, "%s: unsupported C=%ld", "xq_perturb_backward", C);
    if (n_pert < 0 || n_pert > B) return xq_set_error(XQ_EINVAL, "%s: n_pert=%ld out of range", "xq_perturb_backward", n_pert);
    const long N = (long)B * HW, Tp = (long)n_pert * HW;
    hipStream_t s = (hipStream_t)stream;
    switch (C) {
        case 8: launch_pbwd<8>(codebook_norm != 0, z, N, HW, Tp, g_out, g_z, g_zq, s); break;
        case 16: launch_pbwd<16>(codebook_norm != 0, z, N, HW, Tp, g_out, g_z, g_zq, s); break;
        case 32: launch_pbwd<32>(codebook_norm != 0, z, N, HW, Tp, g_out, g_z, g_zq, s); break;
        case 64: launch_pbwd<64>(codebook_norm != 0, z, N, HW, Tp, g_out, g_z, g_zq, s); break;
    }
    return xq_check_launch("perturb_backward_kernel");
}
