// xq_msvq.hip — multi-scale residual quantizer (VAR-style ladder) on gfx950.
//
// Replaces reference VectorQuantizer2 (tokenizer/tokenizer_image/quant.py; original models/quant.py):
//   forward            quant.py:64-144   (training ladder with quantizer dropout, split vq/commit losses)
//   f_to_idxBl_or_fhat quant.py:182-223  (inference ladder: per-scale indices or cumulative f_hat)
//   Phi.forward        quant.py:261-268  (h*(1-r) + conv3x3(h)*r)
// Per scale s (sequential: scale s+1 needs f_rest after scale s):
//   area-pool f_rest -> pn x pn (quant.py:91)  | nearest code: cosine argmax / raw L2 (xq_vq.hip assign kernel, :93-101)
//   gather E[idx] + bicubic upsample to HxW (:106-109) | Phi (:110-113) | f_hat += h*mask, f_rest -= h (:115-118)
//   masked squared error partial sums for the vq/commit losses (:129-132), code-usage histogram (:102)
// All tensors of the ladder are tiny (B x 32 x 11 x 11): these kernels are launch/latency bound, not HBM bound;
// the arithmetic contract (fma-chain orders) is the one of oracle/xq_oracle.c so results are bit-identical to it.
#include "xq_common.hpp"
#include "xq_internal.hpp"
#include "../../include/xq_ops.h"

#include <math.h>
#include <stdio.h>

using namespace xq;

static constexpr int MS_MAX_HW = 16;   // ladder grids up to 16 x 16 (reference: 11x11 and 16x16)
static constexpr int MS_MAX_SN = 16;

struct Taps {  // bicubic 1-D tap table (passed by value as a kernel argument)
    float w[MS_MAX_HW][4];
    int i0[MS_MAX_HW];
};

// same evaluation as oracle xqo_bicubic_taps: double, rounded once
static void make_taps(int in_size, int out_size, Taps *t) {
    const double A = -0.75;
    const double scale = (double)in_size / (double)out_size;
    for (int o = 0; o < out_size; ++o) {
        double src = scale * ((double)o + 0.5) - 0.5;
        double fl = floor(src);
        double tt = src - fl;
        t->i0[o] = (int)fl - 1;
        double x;
        x = tt + 1.0; t->w[o][0] = (float)(((A * x - 5.0 * A) * x + 8.0 * A) * x - 4.0 * A);
        x = tt;       t->w[o][1] = (float)(((A + 2.0) * x - (A + 3.0)) * x * x + 1.0);
        x = 1.0 - tt; t->w[o][2] = (float)(((A + 2.0) * x - (A + 3.0)) * x * x + 1.0);
        x = 2.0 - tt; t->w[o][3] = (float)(((A * x - 5.0 * A) * x + 8.0 * A) * x - 4.0 * A);
    }
}

// F.interpolate(mode='area') == adaptive_avg_pool2d: window [floor(i*H/ph), ceil((i+1)*H/ph)), row-major
// sequential sum, one division by the window size.  in [BC][H][W] -> out [BC][ph*pw]
__global__ __launch_bounds__(256) void area_pool_kernel(const float *__restrict__ in, long BC, int H, int W, int ph, int pw,
                                                        float *__restrict__ out) {
    const long o = (long)blockIdx.x * 256 + threadIdx.x;
    if (o >= BC * ph * pw) return;
    const long bc = o / (ph * pw);
    const int p = (int)(o - bc * (ph * pw));
    const int i = p / pw, j = p - i * pw;
    const int y0 = (i * H) / ph, y1 = ((i + 1) * H + ph - 1) / ph;
    const int x0 = (j * W) / pw, x1 = ((j + 1) * W + pw - 1) / pw;
    const float *src = in + (size_t)bc * H * W;
    float s = 0.0f;
    for (int y = y0; y < y1; ++y)
        for (int x = x0; x < x1; ++x) s += src[y * W + x];
    out[o] = s / (float)((y1 - y0) * (x1 - x0));
}

// keys -> int64 indices (+ histogram)
__global__ __launch_bounds__(256) void ms_keys_to_idx_kernel(const unsigned long long *__restrict__ keys, long N,
                                                             int64_t *__restrict__ idx, float *__restrict__ hist) {
    const long n = (long)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const unsigned j = (unsigned)(keys[n] & 0xffffffffull);
    idx[n] = (int64_t)j;
    if (hist) atomicAdd(hist + j, 1.0f);
}

// u[b][c][y][x] = bicubic-upsampled gather of E[idx[b][.]][c]  (quant.py:106-109); plain gather on the last scale
__global__ __launch_bounds__(256) void gather_up_kernel(const float *__restrict__ E, int C, const int64_t *__restrict__ idx,
                                                        long B, int pn, int H, int W, int do_bicubic, Taps ty, Taps tx,
                                                        float *__restrict__ u) {
    const long o = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = B * C * H * W;
    if (o >= total) return;
    const int x = (int)(o % W);
    const int y = (int)((o / W) % H);
    const int c = (int)((o / ((long)W * H)) % C);
    const long b = o / ((long)W * H * C);
    const int64_t *ib = idx + b * pn * pn;
    if (!do_bicubic) {
        u[o] = E[(size_t)ib[y * W + x] * C + c];
        return;
    }
    float acc = 0.0f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        int yy = ty.i0[y] + a;
        yy = yy < 0 ? 0 : (yy > pn - 1 ? pn - 1 : yy);
        float r = 0.0f;
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
            int xx = tx.i0[x] + bb;
            xx = xx < 0 ? 0 : (xx > pn - 1 ? pn - 1 : xx);
            r = __builtin_fmaf(tx.w[x][bb], E[(size_t)ib[yy * pn + xx] * C + c], r);
        }
        acc = __builtin_fmaf(ty.w[y][a], r, acc);
    }
    u[o] = acc;
}

// Phi + ladder update + masked loss partial:
//   h = u*(1-r) + (conv3x3(u)+bias)*r ; f_hat += h*mask ; f_rest -= h ; sq += mask*(f_hat - f)^2
// conv = fmaf chain over (ci, ky, kx) ascending starting from the bias, out-of-image taps skipped (oracle xqo_phi).
__global__ __launch_bounds__(256) void phi_update_kernel(const float *__restrict__ u, long B, int C, int H, int W,
                                                         const float *__restrict__ wgt, const float *__restrict__ bias,
                                                         float ratio, int has_phi, const float *__restrict__ n_quant, int si,
                                                         const float *__restrict__ f, float *__restrict__ f_hat,
                                                         float *__restrict__ f_rest, float *__restrict__ h_out,
                                                         float *__restrict__ fhat_scale_out, double *__restrict__ sq_acc) {
    const long o = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = B * C * H * W;
    float contrib = 0.0f;
    if (o < total) {
        const int x = (int)(o % W);
        const int y = (int)((o / W) % H);
        const int co = (int)((o / ((long)W * H)) % C);
        const long b = o / ((long)W * H * C);
        float h = u[o];
        if (has_phi) {
            float acc = bias[co];
            const float *ub = u + (size_t)b * C * H * W;
            const float *wc = wgt + (size_t)co * C * 9;
            for (int ci = 0; ci < C; ++ci) {
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int yy = y + ky - 1;
                    if (yy < 0 || yy >= H) continue;
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const int xx = x + kx - 1;
                        if (xx < 0 || xx >= W) continue;
                        acc = __builtin_fmaf(wc[ci * 9 + ky * 3 + kx], ub[((size_t)ci * H + yy) * W + xx], acc);
                    }
                }
            }
            h = h * (1.0f - ratio) + acc * ratio;
        }
        const float m = (n_quant == nullptr || (float)si < n_quant[b]) ? 1.0f : 0.0f;
        const float fh = f_hat[o] + h * m;
        f_hat[o] = fh;
        if (f_rest) f_rest[o] = f_rest[o] - h;
        if (h_out) h_out[o] = h;
        if (fhat_scale_out) fhat_scale_out[o] = fh;
        if (f) {
            const float df = fh - f[o];
            contrib = m * df * df;
        }
    }
    if (sq_acc) {
        __shared__ float red[4];
        float s = wave_sum(contrib);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(sq_acc + si, (double)((red[0] + red[1]) + (red[2] + red[3])));
    }
}

__global__ __launch_bounds__(256) void ms_init_kernel(const float *__restrict__ f, long total, float *__restrict__ f_rest,
                                                      float *__restrict__ f_hat, double *__restrict__ sq_acc, int SN) {
    const long o = (long)blockIdx.x * 256 + threadIdx.x;
    if (o < total) { f_rest[o] = f[o]; f_hat[o] = 0.0f; }
    if (o < SN) sq_acc[o] = 0.0;
}

// f_hat_ste = (f_hat - f) + f (quant.py:135, value of the straight-through form); sq_sum[s] = (float) acc[s]
__global__ __launch_bounds__(256) void ms_final_kernel(const float *__restrict__ f, const float *__restrict__ f_hat, long total,
                                                       float *__restrict__ f_hat_ste, const double *__restrict__ sq_acc,
                                                       float *__restrict__ sq_sum, int SN) {
    const long o = (long)blockIdx.x * 256 + threadIdx.x;
    if (o < total && f_hat_ste) f_hat_ste[o] = (f_hat[o] - f[o]) + f[o];
    if (o < SN && sq_sum) sq_sum[o] = (float)sq_acc[o];
}

// ------------------------------------------------------------------------------------------------
struct MsWs {
    AssignWs aw;
    float *f_rest, *pooled, *u;
    double *sq_acc;
};

static size_t ms_ws_layout(long B, int C, int H, int W, int V, char *base, MsWs *ws) {
    const long total = B * C * H * W;
    AssignWs aw;
    size_t off = assign_ws_layout(B * H * W, C, V, base, &aw);
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t o_rest = take((size_t)total * 4), o_pool = take((size_t)total * 4), o_u = take((size_t)total * 4);
    const size_t o_sq = take(MS_MAX_SN * 8);
    if (ws) {
        ws->aw = aw;
        ws->f_rest = (float *)(base + o_rest);
        ws->pooled = (float *)(base + o_pool);
        ws->u = (float *)(base + o_u);
        ws->sq_acc = (double *)(base + o_sq);
    }
    return off;
}

extern "C" size_t xq_msvq_workspace_bytes(int B, int C, int H, int W, int V) {
    if (B < 0 || C < 1 || H < 1 || W < 1 || V < 1) return 0;
    return ms_ws_layout(B, C, H, W, V, nullptr, nullptr);
}

// LFQ code of every token: bit c of idx = [tok_c > 0]; tok is [B][C][tHW]
__global__ __launch_bounds__(256) void lfq_sign_idx_kernel(const float *__restrict__ tok, long Ns, int C, int tHW, int bits,
                                                           int64_t *__restrict__ idx, float *__restrict__ hist) {
    const long n = (long)blockIdx.x * 256 + threadIdx.x;
    if (n >= Ns) return;
    const long b = n / tHW, p = n - b * tHW;
    int64_t id = 0;
    for (int c = 0; c < bits; ++c)
        if (tok[((size_t)b * C + c) * tHW + p] > 0.0f) id |= ((int64_t)1 << c);
    idx[n] = id;
    if (hist) atomicAdd(hist + id, 1.0f);
}

extern "C" int xq_msvq_forward(const float *f, int B, int C, int H, int W, const float *E, int V, int using_znorm,
                               const int32_t *patch_nums, int SN, const int32_t *phi_sel, const float *phi_w,
                               const float *phi_b, float phi_ratio, int n_phi, const float *n_quant, int skip_last_pool,
                               int64_t *idx_all, float *f_hat, float *f_hat_ste, float *h_scales, float *u_scales,
                               float *sq_sum, float *hist, float *f_hat_scales, void *workspace, size_t workspace_bytes,
                               xq_stream_t stream) {
    int rc = check_common("xq_msvq_forward", f, B, C, H * W, E, V);
    if (rc) return rc;
    if (B == 0) return XQ_OK;
    if (!patch_nums || SN < 1 || SN > MS_MAX_SN) return xq_set_error(XQ_EINVAL, "%s: need 1 <= SN <= %ld scales", "xq_msvq_forward", MS_MAX_SN);
    if (H > MS_MAX_HW || W > MS_MAX_HW) return xq_set_error(XQ_EINVAL, "%s: grid %ldx%ld exceeds 16x16", "xq_msvq_forward", H, W);
    if (!idx_all || !f_hat) return xq_set_error(XQ_EINVAL, "%s: null idx_all/f_hat", "xq_msvq_forward");
    if (n_phi > 0 && (!phi_sel || !phi_w || !phi_b)) return xq_set_error(XQ_EINVAL, "%s: phi tensors missing", "xq_msvq_forward");
    for (int si = 0; si < SN; ++si) {
        if (patch_nums[si] < 1 || patch_nums[si] > MS_MAX_HW) return xq_set_error(XQ_EINVAL, "%s: bad patch_num at scale %ld", "xq_msvq_forward", si);
        if (n_phi > 0 && (phi_sel[si] < 0 || phi_sel[si] >= n_phi)) return xq_set_error(XQ_EINVAL, "%s: phi_sel out of range at scale %ld", "xq_msvq_forward", si);
    }
    if (patch_nums[SN - 1] * patch_nums[SN - 1] != H * W || H != W)
        return xq_set_error(XQ_EINVAL, "%s: the last scale must equal the (square) latent grid (quant.py:108-109)", "xq_msvq_forward");
    MsWs ws;
    const size_t need = ms_ws_layout(B, C, H, W, V, (char *)workspace, &ws);
    if (!workspace || workspace_bytes < need)
        return xq_set_error(XQ_ENOSPACE, "%s: workspace %ld < %ld bytes", "xq_msvq_forward", (long)workspace_bytes, (long)need);
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)B * C * H * W;
    const unsigned eb = (unsigned)((total + 255) / 256);
    const int mode = using_znorm ? XQ_MODE_COSINE : XQ_MODE_L2_RAW;

    hipLaunchKernelGGL(ms_init_kernel, dim3(eb), dim3(256), 0, s, f, total, ws.f_rest, f_hat, ws.sq_acc, SN);
    // using_znorm == 2: LFQ sign quantisation (lookup_free_quantize.py:182-183): the code is the sign pattern of the pooled
    // residual, idx = sum_c [x_c > 0] << c over the log2(V) bit channels; E holds the +-scale corners in that order
    const bool sign_mode = using_znorm == 2;
    int sign_bits = 0;
    while ((1 << sign_bits) < V) ++sign_bits;
    if (sign_mode && ((1 << sign_bits) != V || sign_bits > C))
        return xq_set_error(XQ_EINVAL, "%s: sign quantisation needs V = 2^bits with bits <= C (V=%ld, C=%ld)", "xq_msvq_forward", V, C);
    if (!sign_mode) {
        rc = launch_assign(mode, C, f, 0, 1, E, V, ws.aw, s, XQI_PREP);  // codebook prepared once for all scales
        if (rc) return rc;
    }
    size_t ioff = 0;
    for (int si = 0; si < SN; ++si) {
        const int pn = patch_nums[si];
        const long Ns = (long)B * pn * pn;
        const float *tok = ws.f_rest;
        int tHW = H * W;
        if (!(si == SN - 1 && skip_last_pool)) {
            hipLaunchKernelGGL(area_pool_kernel, dim3((unsigned)(((long)B * C * pn * pn + 255) / 256)), dim3(256), 0, s, ws.f_rest,
                               (long)B * C, H, W, pn, pn, ws.pooled);
            tok = ws.pooled;
            tHW = pn * pn;
        }
        if (sign_mode) {
            hipLaunchKernelGGL(lfq_sign_idx_kernel, dim3((unsigned)((Ns + 255) / 256)), dim3(256), 0, s, tok, Ns, C, tHW, sign_bits,
                               idx_all + ioff, hist ? hist + (size_t)si * V : nullptr);
        } else {
            rc = launch_assign(mode, C, tok, Ns, tHW, E, V, ws.aw, s, XQI_SEARCH);
            if (rc) return rc;
            hipLaunchKernelGGL(ms_keys_to_idx_kernel, dim3((unsigned)((Ns + 255) / 256)), dim3(256), 0, s, ws.aw.keys, Ns, idx_all + ioff,
                               hist ? hist + (size_t)si * V : nullptr);
        }
        Taps ty, tx;
        const int bic = (si != SN - 1);
        if (bic) { make_taps(pn, H, &ty); make_taps(pn, W, &tx); } else { ty = Taps(); tx = Taps(); }
        float *u = u_scales ? u_scales + (size_t)si * total : ws.u;
        hipLaunchKernelGGL(gather_up_kernel, dim3(eb), dim3(256), 0, s, E, C, idx_all + ioff, (long)B, pn, H, W, bic, ty, tx, u);
        const int k = n_phi > 0 ? phi_sel[si] : 0;
        hipLaunchKernelGGL(phi_update_kernel, dim3(eb), dim3(256), 0, s, u, (long)B, C, H, W,
                           n_phi > 0 ? phi_w + (size_t)k * C * C * 9 : nullptr, n_phi > 0 ? phi_b + (size_t)k * C : nullptr, phi_ratio,
                           n_phi > 0 ? 1 : 0, n_quant, si, f, f_hat, ws.f_rest, h_scales ? h_scales + (size_t)si * total : nullptr,
                           f_hat_scales ? f_hat_scales + (size_t)si * total : nullptr, sq_sum ? ws.sq_acc : nullptr);
        rc = xq_check_launch("msvq scale kernels");
        if (rc) return rc;
        ioff += (size_t)Ns;
    }
    hipLaunchKernelGGL(ms_final_kernel, dim3(eb), dim3(256), 0, s, f, f_hat, total, f_hat_ste, ws.sq_acc, sq_sum, SN);
    return xq_check_launch("ms_final_kernel");
}

// ================================================================================================
// backward of VectorQuantizer2.forward (autograd-derived upstream; SURVEY.md §8a "Backward structure")
//   out = sg(f_hat - f) + f                      -> g_f += g_out
//   sq_commit[s] = sum m_s (sg(f_hat_s) - f)^2  (commit numerators, :132) -> g_f  += 2 g_sq_commit[s] m_s (f - f_hat_s)
//   sq_vq[s]     = sum m_s (f_hat_s - sg f)^2   (vq numerators, :131)     -> G_s   = 2 g_sq_vq[s] m_s (f_hat_s - f)
//   (the host mirror forms mean_vq_loss / mean_commit_loss from the two numerator vectors with ordinary tensor ops,
//    so beta, 1/ratio_s, 1/SN and the models/quant.py variant all arrive here through g_sq_vq / g_sq_commit)
//   f_hat_s = sum_{t<=s} m_t h_t  =>  dL/dh_t = m_t sum_{s>=t} G_s ;  h_t = Phi_k(t)(u_t), u_t = up(E[idx_t])
// ================================================================================================
struct TapsT { float w[MS_MAX_HW][MS_MAX_HW]; };  // transposed dense 1-D interpolation matrix [src][dst]

static void make_taps_t(int in_size, int out_size, TapsT *t) {
    Taps tp;
    make_taps(in_size, out_size, &tp);
    for (int i = 0; i < MS_MAX_HW; ++i)
        for (int o = 0; o < MS_MAX_HW; ++o) t->w[i][o] = 0.0f;
    for (int o = 0; o < out_size; ++o)
        for (int a = 0; a < 4; ++a) {
            int ii = tp.i0[o] + a;
            ii = ii < 0 ? 0 : (ii > in_size - 1 ? in_size - 1 : ii);
            t->w[ii][o] += tp.w[o][a];
        }
}

__global__ __launch_bounds__(256) void ms_bwd_elem_kernel(const float *__restrict__ f, const float *__restrict__ h_scales,
                                                          long total, long per_sample, int SN, const float *__restrict__ n_quant,
                                                          const float *__restrict__ g_out, const float *__restrict__ g_sq_vq,
                                                          const float *__restrict__ g_sq_commit, float *__restrict__ g_f,
                                                          float *__restrict__ gh) {
    const long o = (long)blockIdx.x * 256 + threadIdx.x;
    if (o >= total) return;
    const long b = o / per_sample;
    const float nq = n_quant ? n_quant[b] : (float)(SN + 1);
    const float fv = f[o];
    float G[MS_MAX_SN];
    float fh = 0.0f, gf = g_out ? g_out[o] : 0.0f;
#pragma unroll
    for (int s = 0; s < MS_MAX_SN; ++s) {
        G[s] = 0.0f;
        if (s < SN) {
            const float m = ((float)s < nq) ? 1.0f : 0.0f;
            fh = fh + h_scales[(size_t)s * total + o] * m;
            const float diff = m * (fh - fv);
            G[s] = 2.0f * (g_sq_vq ? g_sq_vq[s] : 0.0f) * diff;          // d sq_vq[s] / d f_hat_s
            gf -= 2.0f * (g_sq_commit ? g_sq_commit[s] : 0.0f) * diff;    // d sq_commit[s] / d f
        }
    }
    g_f[o] = gf;
    float run = 0.0f;
#pragma unroll
    for (int t = MS_MAX_SN - 1; t >= 0; --t) {
        if (t < SN) {
            run += G[t];
            const float m = ((float)t < nq) ? 1.0f : 0.0f;
            gh[(size_t)t * total + o] = m * run;
        }
    }
}

// g_u = (1-r) g_h + r conv3x3^T(g_h, W)
__global__ __launch_bounds__(256) void phi_bwd_input_kernel(const float *__restrict__ gh, long B, int C, int H, int W,
                                                            const float *__restrict__ wgt, float ratio, float *__restrict__ gu) {
    const long o = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = B * C * H * W;
    if (o >= total) return;
    const int x = (int)(o % W);
    const int y = (int)((o / W) % H);
    const int ci = (int)((o / ((long)W * H)) % C);
    const long b = o / ((long)W * H * C);
    const float *gb = gh + (size_t)b * C * H * W;
    float acc = 0.0f;
    for (int co = 0; co < C; ++co) {
        const float *wc = wgt + ((size_t)co * C + ci) * 9;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int yy = y - ky + 1;
            if (yy < 0 || yy >= H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int xx = x - kx + 1;
                if (xx < 0 || xx >= W) continue;
                acc = __builtin_fmaf(wc[ky * 3 + kx], gb[((size_t)co * H + yy) * W + xx], acc);
            }
        }
    }
    gu[o] = gh[o] * (1.0f - ratio) + acc * ratio;
}

// one block per (co, ci): gW[co][ci][ky][kx] += r * sum_{b,y,x} g_h[b][co][y][x] * u[b][ci][y+ky-1][x+kx-1];
// blocks with ci == 0 also add the bias grad gb[co] += r * sum g_h[b][co][.][.]
__global__ __launch_bounds__(256) void phi_bwd_weight_kernel(const float *__restrict__ gh, const float *__restrict__ u, long B,
                                                             int C, int H, int W, float ratio, float *__restrict__ gW,
                                                             float *__restrict__ gb) {
    const int co = blockIdx.x / C, ci = blockIdx.x % C;
    float acc[10];
#pragma unroll
    for (int i = 0; i < 10; ++i) acc[i] = 0.0f;
    const long n = B * H * W;
    for (long e = threadIdx.x; e < n; e += 256) {
        const int x = (int)(e % W);
        const int y = (int)((e / W) % H);
        const long b = e / ((long)W * H);
        const float g = gh[(((size_t)b * C + co) * H + y) * W + x];
        const float *ub = u + ((size_t)b * C + ci) * H * W;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int yy = y + ky - 1;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int xx = x + kx - 1;
                const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
                acc[ky * 3 + kx] = __builtin_fmaf(g, ok ? ub[yy * W + xx] : 0.0f, acc[ky * 3 + kx]);
            }
        }
        acc[9] += g;
    }
    __shared__ float red[4][10];
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const float s = wave_sum(acc[i]);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][i] = s;
    }
    __syncthreads();
    if (threadIdx.x < 10) {
        const int i = threadIdx.x;
        const float s = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
        if (i < 9) gW[((size_t)co * C + ci) * 9 + i] += ratio * s;
        else if (ci == 0) gb[co] += ratio * s;
    }
}

// g_E[idx[b][p]][c] += sum_{y,x} Wy[py][y] Wx[px][x] g_u[b][c][y][x]   (transpose of the bicubic upsample + gather)
__global__ __launch_bounds__(256) void up_bwd_scatter_kernel(const float *__restrict__ gu, long B, int C, int H, int W, int pn,
                                                             int do_bicubic, TapsT ty, TapsT tx, const int64_t *__restrict__ idx,
                                                             float *__restrict__ gE) {
    const long o = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = B * C * pn * pn;
    if (o >= total) return;
    const int px = (int)(o % pn);
    const int py = (int)((o / pn) % pn);
    const int c = (int)((o / ((long)pn * pn)) % C);
    const long b = o / ((long)pn * pn * C);
    const float *g = gu + ((size_t)b * C + c) * H * W;
    float val;
    if (!do_bicubic) {
        val = g[py * W + px];
    } else {
        val = 0.0f;
        for (int y = 0; y < H; ++y) {
            const float wy = ty.w[py][y];
            if (wy == 0.0f) continue;
            float r = 0.0f;
            for (int x = 0; x < W; ++x) r = __builtin_fmaf(tx.w[px][x], g[y * W + x], r);
            val = __builtin_fmaf(wy, r, val);
        }
    }
    atomicAdd(gE + (size_t)idx[b * pn * pn + py * pn + px] * C + c, val);
}

extern "C" size_t xq_msvq_backward_workspace_bytes(int B, int C, int H, int W, int SN) {
    if (B < 0 || C < 1 || H < 1 || W < 1 || SN < 1) return 0;
    const size_t total = (size_t)B * C * H * W;
    return align_up(total * 4 * (size_t)SN, 256) + align_up(total * 4, 256);
}

extern "C" int xq_msvq_backward(const float *f, int B, int C, int H, int W, int V, const int32_t *patch_nums, int SN,
                                const int32_t *phi_sel, const float *phi_w, float phi_ratio, int n_phi, const float *n_quant,
                                const int64_t *idx_all, const float *h_scales,
                                const float *u_scales, const float *g_out, const float *g_sq_vq, const float *g_sq_commit, float *g_f,
                                float *g_E, float *g_phi_w, float *g_phi_b, void *workspace, size_t workspace_bytes,
                                xq_stream_t stream) {
    if (B == 0) return XQ_OK;
    if (!f || !patch_nums || !idx_all || !h_scales || !g_f || !g_E)
        return xq_set_error(XQ_EINVAL, "%s: null pointer", "xq_msvq_backward");
    if (SN < 1 || SN > MS_MAX_SN || H > MS_MAX_HW || W > MS_MAX_HW || V < 1)
        return xq_set_error(XQ_EINVAL, "%s: unsupported ladder (SN=%ld, H=%ld)", "xq_msvq_backward", SN, H);
    if (n_phi > 0 && (!phi_sel || !phi_w || !u_scales || !g_phi_w || !g_phi_b))
        return xq_set_error(XQ_EINVAL, "%s: phi tensors missing", "xq_msvq_backward");
    const size_t need = xq_msvq_backward_workspace_bytes(B, C, H, W, SN);
    if (!workspace || workspace_bytes < need)
        return xq_set_error(XQ_ENOSPACE, "%s: workspace %ld < %ld bytes", "xq_msvq_backward", (long)workspace_bytes, (long)need);
    hipStream_t s = (hipStream_t)stream;
    const long total = (long)B * C * H * W;
    const unsigned eb = (unsigned)((total + 255) / 256);
    float *gh = (float *)workspace;
    float *gu = (float *)((char *)workspace + align_up((size_t)total * 4 * SN, 256));
    hipLaunchKernelGGL(ms_bwd_elem_kernel, dim3(eb), dim3(256), 0, s, f, h_scales, total, (long)C * H * W, SN, n_quant, g_out, g_sq_vq,
                       g_sq_commit, g_f, gh);
    int rc = xq_check_launch("ms_bwd_elem_kernel");
    if (rc) return rc;
    size_t ioff = 0;
    for (int si = 0; si < SN; ++si) {
        const int pn = patch_nums[si];
        const float *ghs = gh + (size_t)si * total;
        const float *gus = ghs;
        if (n_phi > 0) {
            const int k = phi_sel[si];
            const float *wk = phi_w + (size_t)k * C * C * 9;
            hipLaunchKernelGGL(phi_bwd_input_kernel, dim3(eb), dim3(256), 0, s, ghs, (long)B, C, H, W, wk, phi_ratio, gu);
            hipLaunchKernelGGL(phi_bwd_weight_kernel, dim3(C * C), dim3(256), 0, s, ghs, u_scales + (size_t)si * total, (long)B, C, H, W,
                               phi_ratio, g_phi_w + (size_t)k * C * C * 9, g_phi_b + (size_t)k * C);
            gus = gu;
        }
        const int bic = (si != SN - 1);
        TapsT ty, tx;
        if (bic) { make_taps_t(pn, H, &ty); make_taps_t(pn, W, &tx); } else { ty = TapsT(); tx = TapsT(); }
        hipLaunchKernelGGL(up_bwd_scatter_kernel, dim3((unsigned)(((long)B * C * pn * pn + 255) / 256)), dim3(256), 0, s, gus, (long)B, C, H,
                           W, pn, bic, ty, tx, idx_all + ioff, g_E);
        rc = xq_check_launch("msvq backward scale kernels");
        if (rc) return rc;
        ioff += (size_t)B * pn * pn;
    }
    return XQ_OK;
}


// ================================================================================================
// VAR-side helpers of VectorQuantizer2 (quant.py:148-180 embed_to_fhat, :226-258 idxBl_to_var_input /
// get_next_autoregressive_input; models/quant.py:107-215): the three primitives of the ladder as stand-alone ops —
// the same kernels (and fma orders) the fused ladder runs, so the results equal oracle/xq_oracle.c bit for bit.
// ================================================================================================
__global__ __launch_bounds__(256) void up_tensor_kernel(const float *__restrict__ h, long BC, int pn, int H, int W, Taps ty, Taps tx,
                                                        float *__restrict__ u) {
    const long o = (long)blockIdx.x * 256 + threadIdx.x;
    if (o >= BC * H * W) return;
    const int x = (int)(o % W);
    const int y = (int)((o / W) % H);
    const float *src = h + (o / ((long)W * H)) * pn * pn;
    float acc = 0.0f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        int yy = ty.i0[y] + a;
        yy = yy < 0 ? 0 : (yy > pn - 1 ? pn - 1 : yy);
        float r = 0.0f;
#pragma unroll
        for (int bb = 0; bb < 4; ++bb) {
            int xx = tx.i0[x] + bb;
            xx = xx < 0 ? 0 : (xx > pn - 1 ? pn - 1 : xx);
            r = __builtin_fmaf(tx.w[x][bb], src[yy * pn + xx], r);
        }
        acc = __builtin_fmaf(ty.w[y][a], r, acc);
    }
    u[o] = acc;
}

static int ms_check_grid(const char *fn, int B, int C, int H, int W) {
    if (B < 0 || C < 1 || H < 1 || W < 1) return xq_set_error(XQ_EINVAL, "%s: bad shape", fn);
    if (H > MS_MAX_HW || W > MS_MAX_HW) return xq_set_error(XQ_EINVAL, "%s: grid %ldx%ld exceeds 16x16", fn, H, W);
    return XQ_OK;
}

extern "C" int xq_ms_upsample(const float *h, const float *E, const int64_t *idx, int B, int C, int pn, int H, int W, int bicubic,
                              float *u, xq_stream_t stream) {
    const char *fn = "xq_ms_upsample";
    int rc = ms_check_grid(fn, B, C, H, W);
    if (rc) return rc;
    if (B == 0) return XQ_OK;
    if (pn < 1 || pn > MS_MAX_HW) return xq_set_error(XQ_EINVAL, "%s: bad patch_num %ld", fn, pn);
    if (!u || (!h && (!E || !idx))) return xq_set_error(XQ_EINVAL, "%s: null pointer", fn);
    if (!bicubic && (pn != H || pn != W)) return xq_set_error(XQ_EINVAL, "%s: without interpolation the source grid must equal the target", fn);
    Taps ty, tx;
    if (bicubic) { make_taps(pn, H, &ty); make_taps(pn, W, &tx); } else { ty = Taps(); tx = Taps(); }
    const long total = (long)B * C * H * W;
    hipStream_t s = (hipStream_t)stream;
    if (h) {
        if (!bicubic) {
            if (hipMemcpyAsync(u, h, (size_t)total * 4, hipMemcpyDeviceToDevice, s) != hipSuccess) return xq_set_error(XQ_ELAUNCH, "%s: copy failed", fn);
            return XQ_OK;
        }
        hipLaunchKernelGGL(up_tensor_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, h, (long)B * C, pn, H, W, ty, tx, u);
    } else {
        hipLaunchKernelGGL(gather_up_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, E, C, idx, (long)B, pn, H, W, bicubic, ty, tx, u);
    }
    return xq_check_launch(fn);
}

extern "C" int xq_ms_phi_accumulate(const float *u, int B, int C, int H, int W, const float *phi_w, const float *phi_b, float ratio,
                                    float *f_hat, xq_stream_t stream) {
    const char *fn = "xq_ms_phi_accumulate";
    int rc = ms_check_grid(fn, B, C, H, W);
    if (rc) return rc;
    if (B == 0) return XQ_OK;
    if (!u || !f_hat || ((phi_w == nullptr) != (phi_b == nullptr))) return xq_set_error(XQ_EINVAL, "%s: null pointer", fn);
    const long total = (long)B * C * H * W;
    hipLaunchKernelGGL(phi_update_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, u, (long)B, C, H, W, phi_w,
                       phi_b, ratio, phi_w ? 1 : 0, (const float *)nullptr, 0, (const float *)nullptr, f_hat, (float *)nullptr,
                       (float *)nullptr, (float *)nullptr, (double *)nullptr);
    return xq_check_launch(fn);
}

extern "C" int xq_ms_area_pool(const float *in, int B, int C, int H, int W, int pn, float *out, xq_stream_t stream) {
    const char *fn = "xq_ms_area_pool";
    int rc = ms_check_grid(fn, B, C, H, W);
    if (rc) return rc;
    if (B == 0) return XQ_OK;
    if (pn < 1 || pn > MS_MAX_HW || !in || !out) return xq_set_error(XQ_EINVAL, "%s: bad argument", fn);
    const long n = (long)B * C * pn * pn;
    hipLaunchKernelGGL(area_pool_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, (long)B * C, H, W, pn, pn, out);
    return xq_check_launch(fn);
}
