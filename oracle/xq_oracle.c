/*
 * xq_oracle.c — TEST INFRASTRUCTURE ONLY (parity checker + CPU baseline), never shipped.
 *
 * Plain-C restatement of the reference's quantizer hot path (lxa9867/ImageFolder):
 *   VectorQuantizer.forward / f_to_idxBl_or_fhat   tokenizer/tokenizer_image/xqgan_model.py:745-833
 *   add_perturbation                               tokenizer/tokenizer_image/latent_perturbation.py:4-35
 *   VectorQuantizer2.forward / f_to_idxBl_or_fhat  tokenizer/tokenizer_image/quant.py:64-223
 *   Phi.forward                                    tokenizer/tokenizer_image/quant.py:261-268
 *
 * The reference's arithmetic lives in PyTorch ATen (torch==2.4.1 pinned in environment.yml:104):
 * F.normalize, einsum/matmul, argmin/argmax, topk, F.interpolate(area|bicubic), conv2d.  ATen's
 * *summation order* is an implementation detail (MKL sgemm / vectorised reductions); this file
 * restates the same expressions with ONE fixed, documented order so that the HIP kernels can be
 * bit-identical to it:
 *
 *   ARITHMETIC CONTRACT (shared with imagefolder_amd/csrc/*.hip)
 *   (A1) every dot product / sum of squares over the channel axis is a sequential fp32 fmaf chain
 *        in ascending channel order starting from +0.0f   (== the v_mfma_f32_32x32x2_f32 chain);
 *   (A2) l2-normalise(x) = x / max(sqrtf(chain(x,x)), 1e-12f), IEEE division and sqrt
 *        (F.normalize eps=1e-12; xqgan_model.py:753-756);
 *   (A3) squared distance d = fl( fl(|z|^2 + |e|^2) - 2*dot ), norm terms added first, the doubled
 *        dot subtracted last (xqgan_model.py:761-763) — 2*dot is exact;
 *   (A4) argmin/argmax return the LOWEST index among equal extrema (torch CPU semantics);
 *   (A5) loss sums are accumulated in double here (the GPU sums fp32 partials; compared to 1e-6 rel).
 *
 * Pinning: the reference ships no tests/golden vectors for this path so the oracle
 * is pinned against outputs of the *imported reference itself* (oracle/make_golden.py ->
 * tests/golden/ fixtures; tests/test_oracle_golden.py): indices must match except where an fp64
 * re-evaluation shows a sub-ulp tie, floats to 1e-6.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define XQ_EPS 1e-12f

enum { XQ_MODE_L2_NORMED = 0, XQ_MODE_L2_RAW = 1, XQ_MODE_COSINE = 2 };

/* (A1) */
static inline float chain_dot(const float *a, const float *b, int C) {
    float acc = 0.0f;
    for (int k = 0; k < C; ++k) acc = fmaf(a[k], b[k], acc);
    return acc;
}

/* (A2); returns the clamped norm */
static inline float l2norm_row(const float *x, int C, float *y) {
    float n = sqrtf(chain_dot(x, x, C));
    if (!(n > XQ_EPS)) n = XQ_EPS; /* clamp_min(eps); NaN -> eps never happens for finite input */
    for (int k = 0; k < C; ++k) y[k] = x[k] / n;
    return n;
}

void xqo_l2normalize_rows(const float *x, int rows, int C, float *y, float *norm_out) {
    for (int r = 0; r < rows; ++r) {
        float n = l2norm_row(x + (size_t)r * C, C, y + (size_t)r * C);
        if (norm_out) norm_out[r] = n;
    }
}

/* gather token n of a [B][C][HW] tensor into a contiguous C-vector */
static inline void load_token(const float *z, int C, int HW, long n, float *out) {
    long b = n / HW, hw = n % HW;
    const float *base = z + (size_t)b * C * HW + hw;
    for (int k = 0; k < C; ++k) out[k] = base[(size_t)k * HW];
}

/*
 * Nearest-code assignment (K2+K3 of SURVEY §2.4).
 *   z: [B][C][HW] fp32 (NCHW), N = B*HW tokens; E: [V][C] raw codebook.
 *   mode L2_NORMED: argmin_j d(zhat, ehat_j)   (VectorQuantizer, codebook_norm=True; xqgan_model.py:753-766)
 *   mode L2_RAW   : argmin_j d(z, e_j)         (codebook_norm=False; quant.py:96-101)
 *   mode COSINE   : argmax_j zhat . ehat_j     (VectorQuantizer2 using_znorm; quant.py:93-94)
 * best_out (optional): the winning d (or the winning -dot for COSINE).
 */
void xqo_assign(const float *z, long N, int C, int HW, const float *E, int V, int mode,
                int64_t *idx_out, float *best_out) {
    float *Eh = (float *)malloc((size_t)V * C * sizeof(float));
    float *ee = (float *)malloc((size_t)V * sizeof(float));
    /* transposed copy [k][j] so the inner loop over codes vectorises; each code keeps its own chain */
    float *Et = (float *)malloc((size_t)V * C * sizeof(float));
    for (int j = 0; j < V; ++j) {
        if (mode == XQ_MODE_L2_RAW) memcpy(Eh + (size_t)j * C, E + (size_t)j * C, C * sizeof(float));
        else l2norm_row(E + (size_t)j * C, C, Eh + (size_t)j * C);
        ee[j] = chain_dot(Eh + (size_t)j * C, Eh + (size_t)j * C, C);
        for (int k = 0; k < C; ++k) Et[(size_t)k * V + j] = Eh[(size_t)j * C + k];
    }
#pragma omp parallel
    {
        float *zt = (float *)malloc(C * sizeof(float));
        float *zh = (float *)malloc(C * sizeof(float));
        float *dot = (float *)malloc((size_t)V * sizeof(float));
#pragma omp for schedule(static)
        for (long n = 0; n < N; ++n) {
            load_token(z, C, HW, n, zt);
            if (mode == XQ_MODE_L2_RAW) memcpy(zh, zt, C * sizeof(float));
            else l2norm_row(zt, C, zh);
            float zz = chain_dot(zh, zh, C);
            for (int j = 0; j < V; ++j) dot[j] = 0.0f;
            for (int k = 0; k < C; ++k) { /* (A1): per code, ascending-k fmaf chain */
                const float zk = zh[k];
                const float *row = Et + (size_t)k * V;
                for (int j = 0; j < V; ++j) dot[j] = fmaf(zk, row[j], dot[j]);
            }
            float best = INFINITY;
            int64_t bi = 0;
            for (int j = 0; j < V; ++j) {
                float d;
                if (mode == XQ_MODE_COSINE) d = 0.0f - dot[j];
                else { float s = zz + ee[j]; d = s - 2.0f * dot[j]; } /* (A3) */
                if (d < best) { best = d; bi = j; }                   /* (A4) */
            }
            idx_out[n] = bi;
            if (best_out) best_out[n] = best;
        }
        free(zt); free(zh); free(dot);
    }
    free(Eh); free(ee); free(Et);
}

/* full distance rows for selected tokens: d_out[t][j], token list tok[t] (for the top-delta path) */
void xqo_dist_rows(const float *z, int C, int HW, const float *E, int V, int mode,
                   const int64_t *tok, long T, float *d_out) {
    float *Eh = (float *)malloc((size_t)V * C * sizeof(float));
    float *ee = (float *)malloc((size_t)V * sizeof(float));
    for (int j = 0; j < V; ++j) {
        if (mode == XQ_MODE_L2_RAW) memcpy(Eh + (size_t)j * C, E + (size_t)j * C, C * sizeof(float));
        else l2norm_row(E + (size_t)j * C, C, Eh + (size_t)j * C);
        ee[j] = chain_dot(Eh + (size_t)j * C, Eh + (size_t)j * C, C);
    }
#pragma omp parallel for schedule(static)
    for (long t = 0; t < T; ++t) {
        float zt[1024], zh[1024];
        load_token(z, C, HW, tok[t], zt);
        if (mode == XQ_MODE_L2_RAW) memcpy(zh, zt, C * sizeof(float));
        else l2norm_row(zt, C, zh);
        float zz = chain_dot(zh, zh, C);
        for (int j = 0; j < V; ++j) {
            float dt = chain_dot(zh, Eh + (size_t)j * C, C);
            float s = zz + ee[j];
            d_out[(size_t)t * V + j] = (mode == XQ_MODE_COSINE) ? (0.0f - dt) : (s - 2.0f * dt);
        }
    }
    free(Eh); free(ee);
}

/*
 * VectorQuantizer.forward value path (xqgan_model.py:745-799) given the assignment:
 *   zq_out[b][c][hw] = ste ? zhat + (ehat_idx - zhat) : ehat_idx        (:769-771,796-799 | :826-831)
 *   loss_sq  = sum over N*C of (ehat_idx - zhat)^2   (commit = beta*loss_sq/(N*C), vq = loss_sq/(N*C); :792-793)
 *   hist[j] += #tokens assigned to j                 (:774)
 * normed=0 -> zhat=z, ehat=e (codebook_norm=False).
 */
void xqo_vq_finish(const float *z, long N, int C, int HW, const float *E, int V, int normed, int ste,
                   const int64_t *idx, float *zq_out, double *loss_sq, float *hist) {
    double acc = 0.0;
    float zt[1024], zh[1024], eh[1024];
    (void)V;
    for (long n = 0; n < N; ++n) {
        load_token(z, C, HW, n, zt);
        const float *e = E + (size_t)idx[n] * C;
        if (normed) { l2norm_row(zt, C, zh); l2norm_row(e, C, eh); }
        else { memcpy(zh, zt, C * sizeof(float)); memcpy(eh, e, C * sizeof(float)); }
        long b = n / HW, hw = n % HW;
        for (int k = 0; k < C; ++k) {
            float diff = eh[k] - zh[k];
            acc += (double)diff * (double)diff;
            if (zq_out) zq_out[(size_t)b * C * HW + (size_t)k * HW + hw] = ste ? (zh[k] + diff) : eh[k];
        }
        if (hist) hist[idx[n]] += 1.0f;
    }
    if (loss_sq) *loss_sq = acc;
}

/*
 * Hand-derived backward of VectorQuantizer.forward (autograd-derived in the reference):
 *   g_zhat = g_out + g_commit*beta*2*(zhat-ehat)/(N*C)          (straight-through :796 + commit :792)
 *   g_ehat = g_vq*2*(ehat-zhat)/(N*C)                           (vq :793)
 *   g_z    = (g_zhat - zhat*(zhat.g_zhat))/|z|      (normalise Jacobian, :753)
 *   g_E[idx] += (g_ehat - ehat*(ehat.g_ehat))/|e|   (normalise Jacobian of the gathered row, :771)
 * Accumulated in double (reference semantics up to fp32 rounding); normed=0 drops the Jacobians.
 */
void xqo_vq_backward(const float *z, long N, int C, int HW, const float *E, int V, int normed,
                     const int64_t *idx, const float *g_out, float g_vq, float g_commit, float beta,
                     float *g_z, float *g_E) {
    double *gE = (double *)calloc((size_t)V * C, sizeof(double));
    const double inv = 1.0 / ((double)N * (double)C);
    float zt[1024], zh[1024], eh[1024];
    for (long n = 0; n < N; ++n) {
        load_token(z, C, HW, n, zt);
        const float *e = E + (size_t)idx[n] * C;
        float nz = 1.0f, ne = 1.0f;
        if (normed) { nz = l2norm_row(zt, C, zh); ne = l2norm_row(e, C, eh); }
        else { memcpy(zh, zt, C * sizeof(float)); memcpy(eh, e, C * sizeof(float)); }
        long b = n / HW, hw = n % HW;
        double gzh[1024], geh[1024], dz = 0.0, de = 0.0;
        for (int k = 0; k < C; ++k) {
            double go = g_out ? (double)g_out[(size_t)b * C * HW + (size_t)k * HW + hw] : 0.0;
            double diff = (double)zh[k] - (double)eh[k];
            gzh[k] = go + (double)g_commit * beta * 2.0 * diff * inv;
            geh[k] = -(double)g_vq * 2.0 * diff * inv;
            dz += gzh[k] * zh[k];
            de += geh[k] * eh[k];
        }
        for (int k = 0; k < C; ++k) {
            double gz = normed ? (gzh[k] - zh[k] * dz) / nz : gzh[k];
            double ge = normed ? (geh[k] - eh[k] * de) / ne : geh[k];
            g_z[(size_t)b * C * HW + (size_t)k * HW + hw] = (float)gz;
            gE[(size_t)idx[n] * C + k] += ge;
        }
    }
    for (size_t i = 0; i < (size_t)V * C; ++i) g_E[i] = (float)gE[i];
    free(gE);
}

/* float -> uint32 whose unsigned order equals the float order (-inf < ... < -0 < +0 < ... < +inf < NaN+) */
static inline uint32_t f2ord(float f) {
    uint32_t u; memcpy(&u, &f, 4);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

static int cmp_u64(const void *a, const void *b) {
    uint64_t x = *(const uint64_t *)a, y = *(const uint64_t *)b;
    return (x > y) - (x < y);
}

/*
 * rank-r selection over a distance row: the code with the r-th smallest d (0-based), ties broken
 * towards the LOWER index (torch.topk(sorted=True) on CPU was observed ascending-index among equals;
 * tie order is unspecified upstream, so parity on exact ties is "same distance").
 * latent_perturbation.py:20-24
 */
void xqo_select_rank(const float *d_rows, long T, int V, const int32_t *rank, int64_t *idx_out) {
#pragma omp parallel
    {
        uint64_t *keys = (uint64_t *)malloc((size_t)V * sizeof(uint64_t));
#pragma omp for schedule(static)
        for (long t = 0; t < T; ++t) {
            for (int j = 0; j < V; ++j) keys[j] = ((uint64_t)f2ord(d_rows[(size_t)t * V + j]) << 32) | (uint32_t)j;
            qsort(keys, V, sizeof(uint64_t), cmp_u64);
            idx_out[t] = (int64_t)(keys[rank[t]] & 0xffffffffu);
        }
        free(keys);
    }
}

/* ------------------------------------------------------------------------------------------------
 * Multi-scale residual ladder pieces (VectorQuantizer2; quant.py:88-132,182-223)
 * ---------------------------------------------------------------------------------------------- */

/* F.interpolate(mode='area') == adaptive_avg_pool2d: window [floor(i*H/ph), ceil((i+1)*H/ph)),
 * row-major sequential fp32 sum then one division by the window size (quant.py:91).  in: [BC][H][W] */
void xqo_area_pool(const float *in, long BC, int H, int W, int ph, int pw, float *out) {
    for (long bc = 0; bc < BC; ++bc)
        for (int i = 0; i < ph; ++i) {
            int y0 = (i * H) / ph, y1 = ((i + 1) * H + ph - 1) / ph;
            for (int j = 0; j < pw; ++j) {
                int x0 = (j * W) / pw, x1 = ((j + 1) * W + pw - 1) / pw;
                float s = 0.0f;
                for (int y = y0; y < y1; ++y)
                    for (int x = x0; x < x1; ++x) s += in[(size_t)bc * H * W + (size_t)y * W + x];
                out[(size_t)bc * ph * pw + (size_t)i * pw + j] = s / (float)((y1 - y0) * (x1 - x0));
            }
        }
}

/* bicubic (A=-0.75, align_corners=False, border-clamped taps) 1-D tap table for in_size -> out_size:
 * w[o][0..3] weights, i0[o] = floor(src)-1 (unclamped); src = (in/out)*(o+0.5)-0.5   (quant.py:107, ATen
 * upsample_bicubic2d: cubic_convolution1/2 with A=-0.75).  The table is evaluated in double and rounded once
 * to fp32 (ATen evaluates it in fp32 with its own association; both are within 1 ulp of the exact taps). */
void xqo_bicubic_taps(int in_size, int out_size, float *w /*[out][4]*/, int32_t *i0 /*[out]*/) {
    const double A = -0.75;
    const double scale = (double)in_size / (double)out_size;
    for (int o = 0; o < out_size; ++o) {
        double src = scale * ((double)o + 0.5) - 0.5;
        double fl = floor(src);
        double t = src - fl;
        i0[o] = (int32_t)fl - 1;
        double x;
        x = t + 1.0; w[o * 4 + 0] = (float)(((A * x - 5.0 * A) * x + 8.0 * A) * x - 4.0 * A);
        x = t;       w[o * 4 + 1] = (float)(((A + 2.0) * x - (A + 3.0)) * x * x + 1.0);
        x = 1.0 - t; w[o * 4 + 2] = (float)(((A + 2.0) * x - (A + 3.0)) * x * x + 1.0);
        x = 2.0 - t; w[o * 4 + 3] = (float)(((A * x - 5.0 * A) * x + 8.0 * A) * x - 4.0 * A);
    }
}

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* bicubic upsample [BC][ph][pw] -> [BC][H][W]; per output: for each of 4 rows, a 4-tap fmaf chain over x
 * (ascending tap, from 0), then a 4-tap fmaf chain over the row results (ascending tap, from 0). */
void xqo_bicubic_up(const float *in, long BC, int ph, int pw, int H, int W, float *out) {
    float *wy = (float *)malloc(sizeof(float) * 4 * H), *wx = (float *)malloc(sizeof(float) * 4 * W);
    int32_t *iy = (int32_t *)malloc(sizeof(int32_t) * H), *ix = (int32_t *)malloc(sizeof(int32_t) * W);
    xqo_bicubic_taps(ph, H, wy, iy);
    xqo_bicubic_taps(pw, W, wx, ix);
    for (long bc = 0; bc < BC; ++bc) {
        const float *src = in + (size_t)bc * ph * pw;
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                float acc = 0.0f;
                for (int a = 0; a < 4; ++a) {
                    int yy = clampi(iy[y] + a, 0, ph - 1);
                    float r = 0.0f;
                    for (int b = 0; b < 4; ++b) {
                        int xx = clampi(ix[x] + b, 0, pw - 1);
                        r = fmaf(wx[x * 4 + b], src[(size_t)yy * pw + xx], r);
                    }
                    acc = fmaf(wy[y * 4 + a], r, acc);
                }
                out[(size_t)bc * H * W + (size_t)y * W + x] = acc;
            }
    }
    free(wy); free(wx); free(iy); free(ix);
}

/* Phi: out = h*(1-r) + (conv3x3(h)+bias)*r   (quant.py:261-268); conv = fmaf chain over (ci,ky,kx)
 * ascending starting from the bias, zero padding contributes nothing (taps outside are skipped). */
void xqo_phi(const float *h, long B, int C, int H, int W, const float *wgt /*[C][C][3][3]*/,
             const float *bias, float ratio, float *out) {
    const float keep = 1.0f - ratio;
#pragma omp parallel for schedule(static)
    for (long b = 0; b < B; ++b)
        for (int co = 0; co < C; ++co)
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x) {
                    float acc = bias[co];
                    for (int ci = 0; ci < C; ++ci)
                        for (int ky = 0; ky < 3; ++ky) {
                            int yy = y + ky - 1;
                            if (yy < 0 || yy >= H) continue;
                            for (int kx = 0; kx < 3; ++kx) {
                                int xx = x + kx - 1;
                                if (xx < 0 || xx >= W) continue;
                                acc = fmaf(wgt[(((size_t)co * C + ci) * 3 + ky) * 3 + kx],
                                           h[(((size_t)b * C + ci) * H + yy) * W + xx], acc);
                            }
                        }
                    size_t o = (((size_t)b * C + co) * H + y) * W + x;
                    out[o] = h[o] * keep + acc * ratio;
                }
}

/* gather codebook rows into NCHW: out[b][c][p] = E[idx[b*P+p]][c]  (quant.py:106-109) */
void xqo_gather_nchw(const float *E, int C, const int64_t *idx, long B, int P, float *out) {
    for (long b = 0; b < B; ++b)
        for (int p = 0; p < P; ++p) {
            const float *e = E + (size_t)idx[b * P + p] * C;
            for (int c = 0; c < C; ++c) out[((size_t)b * C + c) * P + p] = e[c];
        }
}

/*
 * VectorQuantizer2 forward ladder (quant.py:64-144 train / :182-223 inference).
 *   f: [B][C][H][W]; patch_nums[SN]; phi_sel[SN] index into phi_w/phi_b (K convs, [K][C][C][3][3], [K][C]);
 *   n_quant[B] (float, quant.py:79-86) or NULL (= SN+1 everywhere); skip_last_pool: the reference's
 *   "last scale uses f_rest directly" predicate evaluated by the caller (quant.py:91-92 vs :200-201).
 * Outputs: idx_all (concatenated per scale, B*pn*pn each), f_hat (masked sum, [B][C][H][W]),
 *   vq_sum[SN], commit_sum[SN] = sum over B*C*H*W of mask*(f_hat_s - f)^2 (double), ratio[SN] = mean(mask),
 *   hist [SN][V] (nullable), f_hat_scales (nullable, [SN][B][C][H][W] cumulative unmasked f_hat for inference).
 */
void xqo_msvq_forward(const float *f, long B, int C, int H, int W, const float *E, int V, int using_znorm,
                      const int32_t *patch_nums, int SN, const int32_t *phi_sel, const float *phi_w,
                      const float *phi_b, float phi_ratio, int has_phi, const float *n_quant, int skip_last_pool,
                      int64_t *idx_all, float *f_hat, double *sq_sum, float *ratio, float *hist,
                      float *f_hat_scales) {
    const size_t total = (size_t)B * C * H * W;
    float *f_rest = (float *)malloc(total * sizeof(float));
    float *hbuf = (float *)malloc(total * sizeof(float));
    float *hphi = (float *)malloc(total * sizeof(float));
    float *pooled = (float *)malloc(total * sizeof(float));
    float *gath = (float *)malloc(total * sizeof(float));
    memcpy(f_rest, f, total * sizeof(float));
    memset(f_hat, 0, total * sizeof(float));
    size_t ioff = 0;
    for (int si = 0; si < SN; ++si) {
        const int pn = patch_nums[si];
        const long Ns = B * pn * pn;
        const float *tok = f_rest;
        int tHW = H * W;
        if (!(si == SN - 1 && skip_last_pool)) {
            xqo_area_pool(f_rest, B * C, H, W, pn, pn, pooled);
            tok = pooled; tHW = pn * pn;
        }
        if (using_znorm == 2) {
            /* LFQ (lookup_free_quantize.py:182-183 / :254-268): code = sign pattern of the pooled residual,
               idx = sum_c [x_c > 0] << c over the log2(V) bit channels; E holds the +-scale corners in that order */
            int bits = 0;
            while ((1 << bits) < V) ++bits;
            for (long n = 0; n < Ns; ++n) {
                const long b = n / tHW, p = n % tHW;
                int64_t id = 0;
                for (int c = 0; c < bits && c < C; ++c)
                    if (tok[((size_t)b * C + c) * tHW + p] > 0.0f) id |= ((int64_t)1 << c);
                idx_all[ioff + n] = id;
            }
        } else {
            xqo_assign(tok, Ns, C, tHW, E, V, using_znorm ? XQ_MODE_COSINE : XQ_MODE_L2_RAW, idx_all + ioff, NULL);
        }
        if (hist) for (long n = 0; n < Ns; ++n) hist[(size_t)si * V + idx_all[ioff + n]] += 1.0f;
        xqo_gather_nchw(E, C, idx_all + ioff, B, pn * pn, gath);
        if (si != SN - 1) xqo_bicubic_up(gath, B * C, pn, pn, H, W, hbuf);
        else memcpy(hbuf, gath, total * sizeof(float)); /* last scale: pn*pn == H*W required (quant.py:108-109) */
        const float *h = hbuf;
        if (has_phi) {
            const int k = phi_sel[si];
            xqo_phi(hbuf, B, C, H, W, phi_w + (size_t)k * C * C * 9, phi_b + (size_t)k * C, phi_ratio, hphi);
            h = hphi;
        }
        double sq = 0.0, msum = 0.0;
        for (long b = 0; b < B; ++b) {
            const float m = (n_quant == NULL || (float)si < n_quant[b]) ? 1.0f : 0.0f;
            msum += m;
            for (size_t e = 0; e < (size_t)C * H * W; ++e) {
                size_t o = (size_t)b * C * H * W + e;
                f_hat[o] = f_hat[o] + h[o] * m;       /* quant.py:115-116 */
                f_rest[o] = f_rest[o] - h[o];         /* quant.py:118 */
                float df = f_hat[o] - f[o];
                sq += (double)m * (double)df * (double)df; /* quant.py:131-132 (masked mse numerators) */
            }
        }
        if (sq_sum) sq_sum[si] = sq;
        if (ratio) ratio[si] = (float)(msum / (double)B);
        if (f_hat_scales) memcpy(f_hat_scales + (size_t)si * total, f_hat, total * sizeof(float));
        ioff += (size_t)Ns;
    }
    free(f_rest); free(hbuf); free(hphi); free(pooled); free(gath);
}
