/*
 * xq_ops.h — C-ABI of libxq_ops.so: MI355X (gfx950) kernels for the XQ-GAN tokenizer hot path.
 *
 * The reference (lxa9867/ImageFolder) has no FFI/plugin interface: its seam is Python nn.Module
 * duck-typing over ATen ops.  These entry points are what a binding for that seam
 * needs; each one names the reference code it replaces.  Rules for every function:
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer owned by the caller
 *     (workspace included) unless stated otherwise; kernels never allocate and never synchronise;
 *   - enqueue-only on `stream` (a hipStream_t passed as void*; NULL = the legacy default stream);
 *   - returns 0 on success, a negative XQ_E* code otherwise (xq_last_error() gives the message for
 *     the calling thread); nothing is enqueued when an argument is rejected;
 *   - stateless and re-entrant: the EMA/usage state of the reference modules stays in the host mirror.
 *
 * Tensor layouts follow the reference: feature maps are NCHW fp32 contiguous ([B][C][H*W]),
 * codebooks are [V][C] fp32 row-major, indices are int64 (torch.long).
 */
#ifndef XQ_OPS_H
#define XQ_OPS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XQ_ABI_VERSION 3      /* 2: round 5's entry points (xq_adamw_ema_step_ex, xq_grad_norm_clip, xq_sn_batched_*, xq_token_assemble_*, ...);
                                  3: round 6 (xq_gemm_fused_schedule, XQ_GEMM_DUO / _PDUO, xq_gemm_colpart_rows = 2 * ceil(M / 128) + _rows_written,
                                     xq_gemm_bf16_nt_gelu_bwd, xq_transpose_bf16_batched, xq_conv3x3_pack_weights_batched) */

#define XQ_OK 0
#define XQ_EINVAL (-1)   /* bad shape / null pointer / unsupported size */
#define XQ_ENOSPACE (-2) /* workspace too small */
#define XQ_ELAUNCH (-3)  /* hipLaunch failed (message has the hip error string) */

/* nearest-code search mode */
#define XQ_MODE_L2_NORMED 0 /* argmin ||zhat-ehat||^2, both l2-normalised: VectorQuantizer(codebook_norm=True), xqgan_model.py:753-766 */
#define XQ_MODE_L2_RAW 1    /* argmin ||z-e||^2 on raw vectors: codebook_norm=False / VectorQuantizer2(using_znorm=False), quant.py:96-101 */
#define XQ_MODE_COSINE 2    /* argmax zhat.ehat: VectorQuantizer2(using_znorm=True), quant.py:93-94 */

typedef void *xq_stream_t;

int xq_abi_version(void);
const char *xq_last_error(void);

/* ---- nearest-code assignment (replaces F.normalize + einsum + argmin: xqgan_model.py:753-766, :812-822;
 *      latent_perturbation.py:9-18; quant.py:93-101,202-208) --------------------------------------- */

/* bytes of workspace xq_assign / xq_vq_forward need for N tokens, C channels, V codes */
size_t xq_assign_workspace_bytes(int64_t N, int C, int V);

/*
 * idx[n] = argmin_j score(z_n, E_j) (lowest index among equal scores), n in [0, B*HW).
 *   z [B][C][HW], E [V][C]; idx int64 [B*HW]; best (nullable) fp32 [B*HW] = winning score.
 *   C in {8,16,32,64} (XQ_EINVAL otherwise: the reference configs use 8 ... 64); V >= 1.
 */
int xq_assign(const float *z, int B, int C, int HW, const float *E, int V, int mode, int64_t *idx, float *best,
              void *workspace, size_t workspace_bytes, xq_stream_t stream);

/* ---- VectorQuantizer (single-scale VQ; xqgan_model.py:722-833) ---------------------------------- */

/*
 * VectorQuantizer.forward value path (xqgan_model.py:745-799) and its inference twin f_to_idxBl_or_fhat
 * (:803-833).  codebook_norm != 0 selects XQ_MODE_L2_NORMED, else XQ_MODE_L2_RAW.
 *   zq     (nullable) [B][C][HW]: ste ? zhat + (ehat[idx]-zhat) (:796-799) : ehat[idx] (:826-831)
 *   idx    [B*HW] int64 (:766)
 *   hist   (nullable) [V] fp32, ACCUMULATES the bincount of idx (:774)
 *   loss_sq(nullable) [1] fp32 = sum over N*C of (ehat[idx]-zhat)^2; vq_loss = loss_sq/(N*C),
 *          commit_loss = beta*loss_sq/(N*C) (:792-793)
 */
int xq_vq_forward(const float *z, int B, int C, int HW, const float *E, int V, int codebook_norm, int ste,
                  float *zq, int64_t *idx, float *hist, float *loss_sq, void *workspace, size_t workspace_bytes,
                  xq_stream_t stream);

/*
 * Backward of VectorQuantizer.forward (autograd-derived upstream; formulas in SURVEY.md §8a):
 *   g_out (nullable) [B][C][HW] grad of the returned z_q; g_vq, g_commit: DEVICE scalars (nullable = 0),
 *   the upstream grads of vq_loss / commit_loss; beta = commit_loss_beta.
 *   g_z [B][C][HW] and g_E [V][C] are overwritten.  g_E[v] = sum over the tokens that chose code v, formed without
 *   floating-point atomics in a fixed order (ascending token inside 256-token chunks, chunk sums ascending inside four
 *   ranges that meet as ((r0 + r1) + r2) + r3): bit-identical from run to run.  workspace: xq_vq_backward_workspace_bytes
 *   (chunk partial sums + first-of-code table); not needed when g_vq is NULL (g_E = 0 then).
 */
size_t xq_vq_backward_workspace_bytes(int64_t N, int C, int V);
int xq_vq_backward(const float *z, int B, int C, int HW, const float *E, int V, int codebook_norm,
                   const int64_t *idx, const float *g_out, const float *g_vq, const float *g_commit, float beta,
                   float *g_z, float *g_E, void *workspace, size_t workspace_bytes, xq_stream_t stream);

/* ---- RobustTok latent perturbation (latent_perturbation.py:4-35) --------------------------------------- */

size_t xq_perturb_workspace_bytes(int64_t n_pert_tokens, int C, int V);

/*
 * out = where(sample < n_pert, zhat + sg(norm(E[pick]) - zhat), zq_in)     (latent_perturbation.py:26-35)
 *   pick_n = the code with the rank[n]-th smallest distance to token n (0-based; ties -> lower index), i.e.
 *   topk(d, delta, largest=False)[rank[n]] (:20-24).  The caller draws rank (:21-23:
 *   rank = rand > alpha ? 0 : randint(0, delta)) — device int32 [n_pert*HW]; n_pert = int(B*beta) (:32).
 *   z (= encoder latent h), zq_in (= quantizer output), out: [B][C][HW]; sel_idx (nullable) int64 [n_pert*HW].
 */
int xq_perturb_forward(const float *z, const float *zq_in, const float *E, int B, int C, int HW, int V,
                       int codebook_norm, int n_pert, const int32_t *rank, float *out, int64_t *sel_idx,
                       void *workspace, size_t workspace_bytes, xq_stream_t stream);

/* backward of the above: g_z = normalise-Jacobian(g_out) for perturbed samples (else 0); g_zq = g_out for the
 * others (else 0).  No gradient reaches the codebook. */
int xq_perturb_backward(const float *z, int B, int C, int HW, int codebook_norm, int n_pert, const float *g_out,
                        float *g_z, float *g_zq, xq_stream_t stream);

/* ---- VectorQuantizer2: multi-scale residual ladder (tokenizer_image/quant.py:13-258; models/quant.py) ------ */

size_t xq_msvq_workspace_bytes(int B, int C, int H, int W, int V);

/*
 * Forward ladder of VectorQuantizer2.forward (quant.py:64-144) and f_to_idxBl_or_fhat (:182-223).
 *   f [B][C][H][W] (H == W <= 16); E [V][C]; using_znorm -> cosine argmax (:93-94) else raw L2 argmin (:96-101).
 *   patch_nums, phi_sel: HOST int32 arrays [SN] (SN <= 16): scale sizes and, per scale, which of the n_phi residual
 *     convs applies (the reference picks it with numpy in PhiPartiallyShared.__getitem__, quant.py:285-288; that
 *     selection stays in the host mirror).  phi_w [n_phi][C][C][3][3], phi_b [n_phi][C]: device; n_phi = 0 -> identity.
 *   n_quant (nullable) device fp32 [B]: scale si contributes to sample b iff si < n_quant[b] (:79-86,115).
 *   skip_last_pool: the caller's evaluation of "last scale reads f_rest directly" (:91-92 vs the hard-coded 16 at :201).
 * Outputs (device): idx_all int64 [B*sum(pn^2)] (scale-major, then sample, then position); f_hat = masked sum (:116);
 *   f_hat_ste (nullable) = (f_hat - f) + f (:135); h_scales / u_scales (nullable) [SN][B][C][H][W] = per-scale
 *   post-/pre-Phi maps kept for the backward; sq_sum (nullable) [SN] = sum mask*(f_hat_s - f)^2 (:131-132 numerators);
 *   hist (nullable) [SN][V] ACCUMULATES bincounts (:102); f_hat_scales (nullable) [SN][B][C][H][W] cumulative f_hat (:221).
 */
/* using_znorm: 0 = nearest code in raw L2 (quant.py:96-101), 1 = cosine (:93-94), 2 = LFQ sign quantisation
 * (lookup_free_quantize.py:182-183, :254-268): idx = sum_c [x_c > 0] << c over log2(V) bit channels, E = the +-scale corners. */
int xq_msvq_forward(const float *f, int B, int C, int H, int W, const float *E, int V, int using_znorm,
                    const int32_t *patch_nums, int SN, const int32_t *phi_sel, const float *phi_w, const float *phi_b,
                    float phi_ratio, int n_phi, const float *n_quant, int skip_last_pool, int64_t *idx_all, float *f_hat,
                    float *f_hat_ste, float *h_scales, float *u_scales, float *sq_sum, float *hist, float *f_hat_scales,
                    void *workspace, size_t workspace_bytes, xq_stream_t stream);

size_t xq_msvq_backward_workspace_bytes(int B, int C, int H, int W, int SN);

/*
 * Backward of VectorQuantizer2.forward (formulas: SURVEY.md §8a).  The forward returns the loss NUMERATORS
 * sq_sum[s]; the host forms mean_vq_loss / mean_commit_loss from them (quant.py:129-134; models/quant.py:95), so the
 * upstream grads arrive per scale: g_sq_vq[s] = dL/d(sum m (f_hat_s - sg f)^2) flows into f_hat_s (hence into
 * E and Phi), g_sq_commit[s] = dL/d(sum m (sg f_hat_s - f)^2) flows into f.  Both are DEVICE fp32 [SN], nullable.
 * g_out (nullable): grad of the returned f_hat (identity to f, :135).  g_f is overwritten; g_E [V][C], g_phi_w,
 * g_phi_b are ACCUMULATED into (caller zeroes them).
 */
int xq_msvq_backward(const float *f, int B, int C, int H, int W, int V, const int32_t *patch_nums, int SN,
                     const int32_t *phi_sel, const float *phi_w, float phi_ratio, int n_phi, const float *n_quant,
                     const int64_t *idx_all, const float *h_scales, const float *u_scales, const float *g_out,
                     const float *g_sq_vq, const float *g_sq_commit, float *g_f, float *g_E, float *g_phi_w,
                     float *g_phi_b, void *workspace, size_t workspace_bytes, xq_stream_t stream);

/* ---- fused optimizer step over flat fp32 arenas (xqgan_train.py:344-347,447,459-462; utils/ema.py:5-14) ---------- */

/*
 * One pass over n parameters: AdamW (torch.optim.AdamW single-tensor semantics, amsgrad=False, maximize=False) on
 * p with gradient g*grad_scale (grad_scale = 1/world_size folds DDP's mean), first/second moments m/v, bias
 * correction for the 1-based `step`; then ema = ema*ema_decay + p*(1-ema_decay) (ema nullable); g is zeroed when
 * zero_grad != 0; p_bf16 (nullable, bf16 [n]) receives a bf16 shadow copy of the updated p (read by the GEMMs
 * instead of re-casting the fp32 masters every step).  All arenas: device, 16-byte aligned, n elements.
 */
int xq_adamw_ema_step(float *p, float *g, float *m, float *v, float *ema, void *p_bf16, int64_t n, float lr, float beta1,
                      float beta2, float eps, float weight_decay, int64_t step, float ema_decay, float grad_scale,
                      int zero_grad, xq_stream_t stream);
/* The same step with the bias-correction factors read from device memory: coeffs [2] = { lr / (1 - beta1^step), 1 / sqrt(1 - beta2^step) }
 * computed by the caller from a device-resident step counter, so that the launch can be recorded in a hipGraph (train.CapturedStep) and
 * still see the step advance on every replay. */
int xq_adamw_ema_step_dev(float *p, float *g, float *m, float *v, float *ema, void *p_bf16, int64_t n, float lr, float beta1,
                          float beta2, float eps, float weight_decay, const float *coeffs, float ema_decay, float grad_scale,
                          int zero_grad, xq_stream_t stream);

/*
 * Gradient clipping by global norm without a host read (xqgan_train.py:456-458,471-473: scaler.unscale_ + torch.nn.utils.clip_grad_norm_
 * (parameters, max_grad_norm) when max_grad_norm != 0; norm_type 2, error_if_nonfinite False).  xq_grad_norm_clip: out2[0] = the 2-norm of
 * g * grad_scale over the flat gradient arena (grad_scale = 1/world_size: DDP averages before the trainer clips), out2[1] =
 * min(1, max_norm / (out2[0] + 1e-6)) (max_norm <= 0: 1).  Two launches, fixed summation order (deterministic), double accumulation.
 * workspace: xq_grad_norm_workspace_bytes() bytes, 8-byte aligned.  xq_adamw_ema_step_ex: xq_adamw_ema_step / _dev in one entry point —
 * `coeffs` (nullable) as in _dev, else the 1-based `step`; `clip2` (nullable) = the out2 above: the step uses g * grad_scale * clip2[1].
 */
size_t xq_grad_norm_workspace_bytes(void);
int xq_grad_norm_clip(const float *g, int64_t n, float grad_scale, float max_norm, void *workspace, size_t workspace_bytes, float *out2,
                      xq_stream_t stream);
int xq_adamw_ema_step_ex(float *p, float *g, float *m, float *v, float *ema, void *p_bf16, int64_t n, float lr, float beta1, float beta2,
                         float eps, float weight_decay, int64_t step, const float *coeffs, const float *clip2, float ema_decay,
                         float grad_scale, int zero_grad, xq_stream_t stream);
/* Transposed bf16 copies of many matrices in one launch (round 6): the [in][out] shadows of the Linear weights that the data gradients read as
 * the K-major operand of an NT product (xq_gemm_bf16_nt on w_t; xq_gemm_bf16_nt_gelu_bwd) instead of transpose-reading W [out][in]
 * (xq_gemm_bf16_nn) — the reference computes the same g_x = g_y W inside autograd's mm backward (torch.nn.Linear; vision_transformer.py:295-339).
 * table: DEVICE int64 [n_mats][5] = {source offset, destination offset (bf16 elements from src / dst), rows, cols, first tile}; rows and cols
 * multiples of 64; a matrix owns (rows / 64) * (cols / 64) consecutive tiles starting at its first tile, `tiles` = their total. */
int xq_transpose_bf16_batched(const void *src, void *dst, const int64_t *table, int n_mats, int64_t tiles, xq_stream_t stream);

/* ---- fused row kernels of the ViT blocks (dino_enc/vision_transformer.py:280-339; timm Mlp) ------------------------
 * Activations are [rows][D] row-major; act_bf16 selects their dtype (1 = bf16, 0 = fp32); the residual stream,
 * LayerNorm statistics and all parameter tensors are fp32 (what bf16 autocast does upstream). */

/* rows of the `partials` workspace of the backward kernels: the per-block partial rows they write + the few rows of the second
 * level of the column sums that follow them (size the workspace with it: rows x quantities x width floats) */
int xq_row_partials_blocks(int64_t rows);

/* Token assembly in front of a block stack (round 5; dino_enc/dinov2.py:149-179,313-349, vision_transformer.py:818-851):
 * out[b][t] = table[t] + (start <= t < start + n ? data[b][t - start] : 0), table fp32 [N][D] = the sample-independent part (class token,
 * position table, learnable latent / mask tokens, level embedding), data fp32 / bf16 [B][n][D] = the per-sample tokens; round_bf16: the value
 * is rounded to bf16 (upstream's cast to the autocast dtype) and kept in fp32.  backward: g fp32 [B][N][D] -> g_data (nullable, data's dtype)
 * = the slice, g_table (nullable) [N][D] = sum over b in ascending order. */
int xq_token_assemble_forward(const float *table, const void *data, int data_bf16, int B, int N, int n, int start, int D, int round_bf16,
                              float *out, xq_stream_t stream);
int xq_token_assemble_backward(const float *g, int data_bf16, int B, int N, int n, int start, int D, void *g_data, float *g_table,
                               xq_stream_t stream);

/* x_new = x + mask[row / rows_per_sample] * (gamma * y)   (LayerScale :291, DropPath, residual :337-338; y/gamma/mask
 * nullable); a = LayerNorm(x_new; lnw, lnb, eps) (:310,323; final norm :959).  x_new (nullable when y is null),
 * mean/rstd [rows] fp32 are saved for the backward.  D in {64,128,256,384,512,768,1024}. */
int xq_res_ln_forward(const float *x, const void *y, const float *gamma, const float *mask, int64_t rows, int D,
                      int rows_per_sample, const float *lnw, const float *lnb, float eps, int act_bf16, float *x_new,
                      void *a, float *mean, float *rstd, xq_stream_t stream);

/* transpose of the above.  g_a: grad of a (nullable); g_xnew: grad arriving on the residual stream (nullable).
 * Outputs: g_x [rows][D] fp32 (grad of x), g_y (dtype of y; required when y is given), and the parameter grads
 * g_lnw, g_lnb, g_gamma, g_ybias (= column sums of g_y: the bias grad of the Linear that produced y), each [D]
 * fp32, nullable, overwritten or accumulated.  partials: workspace of xq_row_partials_blocks(rows)*4*D floats. */
int xq_res_ln_backward(const void *g_a, const float *g_xnew, const float *x_new, const float *mean, const float *rstd,
                       const float *lnw, const void *y, const float *gamma, const float *mask, int64_t rows, int D,
                       int rows_per_sample, int act_bf16, float *g_x, void *g_y, float *g_lnw, float *g_lnb,
                       float *g_gamma, float *g_ybias, int accumulate, float *partials, xq_stream_t stream);

/* GELU: approximate_tanh = 0: exact erf form, nn.GELU() of timm's Mlp (vision_transformer.py Mlp.act);
 * approximate_tanh = 1: F.gelu(approximate='tanh') of the discriminator's frozen DINO blocks (discriminator_dino.py:37-112).
 * n elements (multiple of the 16-byte vector) */
int xq_gelu_forward(const void *h, int64_t n, int act_bf16, int approximate_tanh, void *out, xq_stream_t stream);
/* g_h = g_out * gelu'(h); g_bias (nullable) [H] = column sums of g_h (the fc1 bias gradient);
 * partials: xq_row_partials_blocks(rows*4)*H floats */
int xq_gelu_backward(const void *g_out, const void *h, int64_t rows, int H, int act_bf16, int approximate_tanh, void *g_h,
                     float *g_bias, int accumulate, float *partials, xq_stream_t stream);
/* out[H] (+)= column sums of g [rows][H] (bias gradient of a Linear); partials as above */
int xq_colsum(const void *g, int64_t rows, int H, int act_bf16, float *out, int accumulate, float *partials,
              xq_stream_t stream);

/* ---- LPIPS feature comparison (lpips.py:85-96,159-164), one fused pass per VGG level -------------------------------
 * val[b] = mean_{h,w} sum_c w_c (f0/(|f0|+1e-10) - f1/(|f1|+1e-10))^2 with f0, f1 channels-last activations
 * ([B][HW][C] contiguous; act_bf16 selects bf16 / fp32), w [C] fp32, val [B] fp32.  C in {64,128,256,512}. */
int xq_lpips_level_forward(const void *f0, const void *f1, const float *w, int B, int HW, int C, int act_bf16, float *val,
                           xq_stream_t stream);
/* g1 [B][HW][C] (dtype of f1) = gout[b] * d val[b] / d f1 */
int xq_lpips_level_backward(const void *f0, const void *f1, const float *w, const float *gout, int B, int HW, int C,
                            int act_bf16, void *g1, xq_stream_t stream);
/* The same gradient inside a hand-driven backward pass of the VGG trunk (lpips.py:118-155: every tapped feature map f1 is the
 * output of a ReLU and also feeds the next slice):  g1 = (gout[b] * d val[b] / d f1 + g_add) * [f1 > 0 if relu_mask]
 * — g_add (nullable, layout and dtype of f1) is the gradient arriving from the deeper slices; with relu_mask the result is already
 * the gradient w.r.t. the PRE-activation of the convolution that produced f1.  Replaces autograd's add_ + threshold_backward passes. */
int xq_lpips_level_backward_fused(const void *f0, const void *f1, const float *w, const float *gout, const void *g_add, int relu_mask,
                                  int B, int HW, int C, int act_bf16, void *g1, xq_stream_t stream);

/* ---- 3x3 convolution, stride 1, pad 1, NHWC bf16, implicit GEMM on MFMA (xqgan_model.py:454-622 conv3x3 layers;
 *      lpips.py:118-155 VGG16 trunk) ------------------------------------------------------------------------------------ */

/* W [Cout][Cin][3][3] fp32 (device) -> Wp bf16 [Cout][9*Cin] with k = (ky*3+kx)*Cin + c (for_data_grad = 0), or the
 * rotated/transposed pack [Cin][9*Cout] that turns the same kernel into the data-gradient conv (for_data_grad = 1). */
int xq_conv3x3_pack_weights(const float *W, int Cout, int Cin, int for_data_grad, void *Wp, xq_stream_t stream);
/* the same for every registered weight of an arena in one launch (round 6).  table: DEVICE int64 [n_weights][6] = {fp32 source W (device address),
 * forward pack [Cout][9 Cin] or 0, data-gradient pack [Cin rounded up to 64][9 Cout] or 0 (pad rows stay as the caller zeroed them), Cout, Cin,
 * first block}; a weight owns ceil(Cout * Cin * 9 / 256) consecutive blocks, `blocks` = their total. */
int xq_conv3x3_pack_weights_batched(const int64_t *table, int n_weights, int64_t blocks, xq_stream_t stream);

/* Y[b,y,x,n] = act(bias[n] + sum_{ky,kx,c} X[b,y+ky-1,x+kx-1,c] * W[n][c][ky][kx]); X [B][H][W][Cin], Y [B][H][W][Cout]
 * bf16 NHWC (= torch channels_last); bias fp32 [Cout] nullable; relu != 0 fuses ReLU.  Cin % 64 == 0, Cout % 64 == 0.
 * out_mask (nullable): bf16 [B][H][W][Cout]; Y is zeroed where out_mask <= 0 — when the call computes a data gradient, the ReLU of the
 * layer below (the trunk walk of lpips.py:118-155 backwards; autograd's threshold_backward pass) folded into the store.  Accepted
 * where xq_conv3x3_nhwc_bf16_takes_out_mask(Cin, Cout) != 0, XQ_EINVAL otherwise (the caller then masks in a pass of its own). */
int xq_conv3x3_nhwc_bf16_takes_out_mask(int Cin, int Cout);
int xq_conv3x3_nhwc_bf16(const void *X, const void *Wp, const float *bias, int B, int H, int W, int Cin, int Cout, int relu,
                         const void *out_mask, void *Y, xq_stream_t stream);

/* Weight gradient of the same convolution (the trainable CNN encoder/decoder, xqgan_model.py:454-622):
 * dWp fp32 [Cout][9*Cin], k = (ky*3+kx)*Cin + c (the forward's packed order), ACCUMULATED with fp32 atomics — the caller
 * zero-initialises it and permutes to [Cout][Cin][3][3].  X [B][H][W][Cin], dY [B][H][W][Cout] bf16 NHWC.
 * Cin % 128 == 0, Cout % 128 == 0. */
int xq_conv3x3_wgrad_nhwc_bf16(const void *X, const void *dY, int B, int H, int W, int Cin, int Cout, float *dWp,
                               xq_stream_t stream);

/* the same weight gradient for the other 3x3 geometries of the CNN encoder / decoder: X [B][Hi][Wi][Cin], dY [B][Ho][Wo][Cout];
 * stride 2 / pad 0 = Downsample (xqgan_model.py:697-704, its (0,1,0,1) padding is the out-of-image zero), upsample2x = 1: the
 * forward conv ran on the nearest-2x upsampled X (Upsample, :682-686). */
int xq_conv3x3_wgrad_nhwc_bf16_ex(const void *X, const void *dY, int B, int Hi, int Wi, int Ho, int Wo, int Cin, int Cout, int stride, int pad,
                                  int upsample2x, float *dWp, xq_stream_t stream);
/* out [B][Ho][Wo][C] = sum over the 2 x 2 blocks of in [B][2 Ho][2 Wo][C] (bf16): the backward of nearest-2x upsampling */
int xq_sumpool2x2_nhwc_bf16(const void *in, int B, int Ho, int Wo, int C, void *out, xq_stream_t stream);

/* ---- the 3-channel 3x3 convolutions at the ends of the CNN tokenizer / at the mouth of the VGG16 trunk (csrc/xq_convio.hip):
 *      HBM-bound, one thread per pixel, weights through scalar loads; bf16 arithmetic, fp32 accumulation. -------------------------- */
/* y [B][H][W][Cout] bf16 (NHWC) = conv3x3(x planar [B][3][H][W], fp32 or bf16; pad 1) + bias; w_kc fp32 [27][Cout],
 * row k = (ky * 3 + kx) * 3 + ci, values rounded to bf16 by the caller.  Cout 64 or 128.  conv_in (xqgan_model.py:495), VGG conv1_1
 * (lpips.py:118-155), and — on rotated weights — the data gradient of conv_out. */
int xq_conv3x3_from3_forward(const void *x_planar, int x_is_bf16, const float *w_kc, const float *bias, int B, int H, int W, int Cout,
                             int relu, void *y_nhwc, xq_stream_t stream);
/* y planar [B][3][H][W] bf16 = conv3x3(x [B][H][W][C] bf16 NHWC; pad 1) + bias; w_pairs = bf16 [3][9][C] (tap = ky * 3 + kx).
 * conv_out (xqgan_model.py:584) and — on rotated weights — the data gradient of conv_in / VGG conv1_1.  C % 8 == 0. */
int xq_conv3x3_to3_forward(const void *x_nhwc, const void *w_pairs, const float *bias, int B, int H, int W, int C, void *y_planar,
                           xq_stream_t stream);
/* cols [B*H*W][32] bf16: the 27 taps (k as above) of every pixel + 5 zeros; with it the weight gradient of a 3 -> Cout conv is
 * xq_gemm_bf16_tn(g [pixels][Cout], cols). */
int xq_im2col27(const void *x_planar, int x_is_bf16, int B, int H, int W, void *cols, xq_stream_t stream);
/* weight gradient of a C -> 3 conv: partials fp32 [xq_conv3x3_to3_wgrad_blocks(B, H)][3][9][C], to be summed over the first axis
 * (fixed order: deterministic).  g planar [B][3][H][W] (bf16 or fp32).  W even, C % 128 == 0. */
int xq_conv3x3_to3_wgrad_blocks(int B, int H);
int xq_conv3x3_to3_wgrad(const void *x_nhwc, const void *g_planar, int g_is_bf16, int B, int H, int W, int C, float *partials,
                         xq_stream_t stream);

/* row softmax of the AttnBlock scores (xqgan_model.py:652-653): p = softmax(bf16(S * scale)) over the N columns, fp32 (P32, kept for
 * the backward) and bf16 (P16, the operand of the second product); backward: dS = bf16(bf16(p * (dP - <p, dP>)) * scale).
 * S, dP, P16, dS bf16 [rows][N]; N % 64 == 0, N <= 1024. */
int xq_row_softmax_forward(const void *S, int64_t rows, int N, float scale, float *P32, void *P16, xq_stream_t stream);
int xq_row_softmax_backward(const float *P32, const void *dP, int64_t rows, int N, float scale, void *dS, xq_stream_t stream);

/* MaxPool2d(kernel_size=2, stride=2) of the VGG16 trunk (lpips.py:118-155), NHWC bf16, even input height/width:
 * X [B][2*Ho][2*Wo][C] -> Y [B][Ho][Wo][C]; the backward recomputes the arg-max from X (first maximum in row-major window
 * order, as ATen) and writes every element of GX [B][2*Ho][2*Wo][C].  C % 8 == 0. */
int xq_maxpool2x2_nhwc_bf16_forward(const void *X, int B, int Ho, int Wo, int C, void *Y, xq_stream_t stream);
int xq_maxpool2x2_nhwc_bf16_backward(const void *X, const void *G, int B, int Ho, int Wo, int C, void *GX, xq_stream_t stream);

/* ---- DiffAug (diffaug.py:64-118: translation, colour, cut-out) on planar fp32 images [B][3][H][W], two launches per direction ------
 * rand01 [7][B]: the call's uniform draws (translation h / w, brightness, saturation, contrast, cut-out h / w) exactly as upstream draws
 * them; dh, dw = round(H / 8), round(W / 8); ch, cw = round(H * cutout), round(W * cutout); trans / color / cut: the three host-side
 * branch decisions.  workspace: xq_diffaug_workspace_floats(B) floats.  backward: gx = (d y / d x)^T g for the same draws. */
size_t xq_diffaug_workspace_floats(int B);
int xq_diffaug_forward(const float *x, const float *rand01, int B, int H, int W, int dh, int dw, int ch, int cw, int trans, int color, int cut,
                       float *y, float *workspace, xq_stream_t stream);
int xq_diffaug_backward(const float *g, const float *rand01, int B, int H, int W, int dh, int dw, int ch, int cw, int trans, int color, int cut,
                        float *gx, float *workspace, xq_stream_t stream);

/* Input side of the frozen DINO-S trunk of the discriminator in one pass per direction (round 5; discriminator_dino.py:327-337 + PatchEmbed
 * :262-276): ImageNet normalisation scale_c * x + shift_c of the [-1, 1] image (scale3 / shift3: HOST pointers, 3 floats), the S x S crop at
 * (oi, oj) (mode 0) or the area resize to S x S (mode 1: F.interpolate(mode = 'area') = adaptive average pooling, windows
 * [floor(o H / S), ceil((o + 1) H / S)); S <= H, W < 2 S), and the patchify of the P x P patch convolution: cols bf16 [(b, gy, gx)][(c, dy, dx)],
 * the A operand of the patch-embedding GEMM.  backward: the GEMM's data gradient gcols (same layout, bf16) -> gx fp32 [B][3][H][W], by gather. */
int xq_dino_prep_patches_forward(const float *x, int B, int H, int W, int S, int P, int mode, int oi, int oj, const float *scale3_host,
                                 const float *shift3_host, void *cols_bf16, xq_stream_t stream);
int xq_dino_prep_patches_backward(const void *gcols_bf16, int B, int H, int W, int S, int P, int mode, int oi, int oj, const float *scale3_host,
                                  const float *shift3_host, float *gx, xq_stream_t stream);

/* out bf16 [B][3][H][W] = scale_c * x + shift_c of an fp32 or bf16 image batch (scale3 / shift3: HOST pointers): the input scaling layer of LPIPS
 * (lpips.py:59-64: (x - shift) / scale) together with autocast's cast in front of the first VGG convolution, one pass; backward
 * gx (fp32 or bf16: the image's dtype) = scale_c * g (g bf16). */
int xq_image_affine_bf16_forward(const void *x, int x_is_bf16, int B, int H, int W, const float *scale3_host, const float *shift3_host,
                                 void *out_bf16, xq_stream_t stream);
int xq_image_affine_bf16_backward(const void *g_bf16, int B, int H, int W, const float *scale3_host, void *gx, int gx_is_bf16, xq_stream_t stream);

/* out [D] = sum over the nrows rows of partials [nrows][D] (fp32), fixed order: finishes the fc1 bias gradient from the column partials of
 * xq_gemm_bf16_nn_gelu_bwd */
int xq_colsum_partials(const float *partials, int nrows, int D, float *out, xq_stream_t stream);

/* ---- the 1-channel logit convolution of a DinoDisc head as a row dot (discriminator_dino.py:215): out[r] = sum_c h[r][c] w[c];
 *      backward: g_h[r][c] = g[r] w[c] (nullable), g_w[c] = sum_r g[r] h[r][c] (nullable; partials: xq_row_partials_blocks(rows * 4) * C floats) */
int xq_rowdot_forward(const void *h, const float *w, int64_t rows, int C, int act_bf16, float *out, xq_stream_t stream);
int xq_rowdot_backward(const void *h, const float *w, const float *g, int64_t rows, int C, int act_bf16, void *g_h, float *g_w,
                       float *partials, xq_stream_t stream);

/* ---- spectral normalisation of the discriminator head convolutions (discriminator_dino.py:121-124: torch spectral_norm, one power
 *      iteration per training forward), fp32 ------------------------------------------------------------------------------------ */
/* out = x / max(|x|_2, eps) (F.normalize); norm_out (nullable) [1] = |x|_2.  One workgroup: n <= a few thousand elements. */
int xq_vec_normalize(const float *x, int n, float eps, float *out, float *norm_out, xq_stream_t stream);
/* out[rows][cols] = g / sigma - (dot / sigma^2) u v^T : the gradient of W / sigma w.r.t. W for sigma = u^T W v with u, v held constant;
 * sigma [1], dot [1] = <g, W> on the device. */
int xq_sn_weight_grad(const float *g, const float *u, const float *v, const float *sigma, const float *dot, int64_t rows, int64_t cols,
                      float *out, xq_stream_t stream);

/* The same, batched over H same-shaped weights (round 5: the five DinoDisc heads hold the same three convolutions; one launch chain per
 * shape instead of one per weight).  W fp32 [H][R][Cin * taps]: Conv1d weights (out, in, taps) flattened as torch's spectral_norm does.
 * forward: one power iteration per weight — u_buf [H][R] / v_buf [H][Cin * taps] (the modules' buffers, stacked) are read and UPDATED in
 * place, u_out / v_out receive the same new vectors (kept for the backward: a later forward moves the buffers on), sigma [H] = u^T W v;
 * out32 fp32 [H][R][taps][Cin] = W / sigma in the reduction order of the unfolded convolution's GEMM (taps = 1: the weight's own layout),
 * out16 (nullable) its bf16 copy.  backward: g fp32 in the layout of out32 -> gW fp32 in the layout of W,
 * gW = g / sigma - (<g, W> / sigma^2) u v^T.  workspace: xq_sn_batched_workspace_floats(H, R, Cin, taps) floats (both calls). */
int xq_sn_batched_workspace_floats(int H, int R, int Cin, int taps);
int xq_sn_batched_forward(const float *W, int H, int R, int Cin, int taps, float eps, float *u_buf, float *v_buf, float *u_out, float *v_out,
                          float *sigma, float *workspace, float *out32, void *out16, xq_stream_t stream);
int xq_sn_batched_backward(const float *g, const float *W, const float *u, const float *v, const float *sigma, int H, int R, int Cin, int taps,
                           float *workspace, float *gW, xq_stream_t stream);

/* ---- multi-head self-attention on the packed qkv projection (dino_enc/vision_transformer.py:175-195: qkv.reshape(B,N,3,H,hd)
 *      .permute(2,0,3,1,4) -> F.scaled_dot_product_attention -> transpose(1,2).reshape(B,N,C); discriminator_dino.py:28).
 *      bf16 MFMA, fp32 softmax statistics, head_dim 64 only, no mask, no dropout. ------------------------------------------ */

/* qkv bf16 [B][N][3][H][64] (the output of the qkv Linear as is); out bf16 [B][N][H*64]; lse fp32 [B][H][N] = natural-log
 * sum-exp of the scaled scores (kept for the backward). */
int xq_attn_forward(const void *qkv, int B, int N, int H, int head_dim, float scale, void *out, float *lse, xq_stream_t stream);

/* dout bf16 [B][N][H*64] -> dqkv bf16 [B][N][3][H][64] (every element written).  delta: fp32 [B][H][N] scratch (round 5: written by the dQ
 * kernel's prologue — dO_i . O_i from the dO fragments it holds anyway — and read by the dK/dV kernel launched behind it; no delta kernel). */
int xq_attn_backward(const void *qkv, const void *out, const void *dout, const float *lse, int B, int N, int H, int head_dim,
                     float scale, void *dqkv, float *delta, xq_stream_t stream);

/* ---- GroupNorm (+ SiLU) of the CNN encoder/decoder (xqgan_model.py:625-672 Normalize + nonlinearity), NHWC bf16 ---------------
 * y = silu?((x - mean_{b,g}) * rstd_{b,g} * w_c + b_c), statistics in fp32 over HW x (C/G) elements (biased variance, two-pass).
 * x, y: bf16 [B][HW][C]; w, b fp32 [C] (nullable); mean, rstd: fp32 [B][G] outputs kept for the backward.
 * Geometry: C % 8 == 0, 256 % (C/8) == 0, C <= 1024, (C/G) % 4 == 0, G <= 64.  workspace: xq_groupnorm_workspace_floats floats. */
size_t xq_groupnorm_workspace_floats(int B, int HW, int C, int G);
int xq_groupnorm_silu_forward(const void *x, const float *w, const float *bias, int B, int HW, int C, int G, float eps, int silu,
                              void *y, float *mean, float *rstd, float *workspace, xq_stream_t stream);
/* dx bf16 [B][HW][C]; g_wb_partials fp32 [rows][2][C] with rows = *n_partial_rows <= B * 64: row-wise partial sums of the bias
 * ([.][0][c]) and weight ([.][1][c]) gradients, to be summed over the first axis by the caller. */
int xq_groupnorm_silu_backward(const void *x, const void *dy, const float *w, const float *bias, const float *mean,
                               const float *rstd, int B, int HW, int C, int G, int silu, void *dx, float *g_wb_partials,
                               int *n_partial_rows, float *workspace, xq_stream_t stream);

/* ---- DinoDisc discriminator heads (discriminator_dino.py:113-166), token-major activations [B][L][C] --------------------
 * BatchNormLocal (:127-154: statistics over virtual batches of 8 samples, i.e. groups of rows_per_group = 8*L consecutive
 * token rows, biased variance, eps inside the sqrt) + LeakyReLU(slope) [+ ResidualBlock: (. + skip) * ratio, :113-119]. */

/* y, skip (nullable), out: [G][rows_per_group][C] (bf16 or fp32 by act_bf16); w, b fp32 [C] (nullable = 1 / 0);
 * mean, rstd: fp32 [G][C] outputs kept for the backward.  C % 64 == 0. */
int xq_bnlocal_lrelu_forward(const void *y, const float *w, const float *b, const void *skip, int G, int rows_per_group, int C,
                             int act_bf16, float eps, float slope, float ratio, void *out, float *mean, float *rstd,
                             xq_stream_t stream);
/* g_y (and g_skip when has_skip) in the activation dtype; gw_part / gb_part: fp32 [G][C] per-group partial sums of the
 * affine gradients (the caller sums them over G). */
int xq_bnlocal_lrelu_backward(const void *g_out, const void *y, const float *w, const float *b, const float *mean, const float *rstd,
                              int G, int rows_per_group, int C, int act_bf16, float slope, float ratio, int has_skip, void *g_y,
                              void *g_skip, float *gw_part, float *gb_part, xq_stream_t stream);
/* class-token readout of the frozen DINO trunk (discriminator_dino.py:339-347: (x[:, 1:] + x[:, :1]) per tapped block): out [B][L][C] (bf16 or
 * fp32) = t[b][l + 1][:] + t[b][0][:] from the fp32 residual stream t [B][L + 1][C]; backward: gt [B][L + 1][C] fp32 with gt[b][l + 1] = g[b][l]
 * and gt[b][0] = sum_l g[b][l] (fixed order).  C % 8 == 0. */
int xq_cls_readout_forward(const float *t, int B, int L, int C, int act_bf16, void *out, xq_stream_t stream);
int xq_cls_readout_backward(const void *g, int B, int L, int C, int act_bf16, float *gt, xq_stream_t stream);
/* im2col of Conv1d(kernel K, padding K/2, padding_mode='circular') (:157-166 make_block with ks = 9):
 * cols[b][l][tap][c] = h[b][(l + tap - K/2) mod L][c]; the conv is then cols[B*L][K*C] x W[Cout][K*C]^T. */
int xq_unfold1d_circular(const void *h, int B, int L, int C, int K, int act_bf16, void *cols, xq_stream_t stream);
/* the transpose: dh[b][l][c] = sum_tap dcols[b][(l - tap + K/2) mod L][tap][c] */
int xq_fold1d_circular(const void *dcols, int B, int L, int C, int K, int act_bf16, void *dh, xq_stream_t stream);

/* ---- VAR-side helpers of VectorQuantizer2 (quant.py:148-180 embed_to_fhat, :226-258 idxBl_to_var_input,
 *      get_next_autoregressive_input): the ladder's three primitives as stand-alone ops (same kernels, same fma orders as
 *      xq_msvq_forward).  fp32 NCHW, grids up to 16 x 16. ---------------------------------------------------------------- */
/* u[B][C][H][W] = F.interpolate(src, (H, W), mode='bicubic') (bicubic = 1) or src itself (bicubic = 0, pn == H == W), where
 * src = h [B][C][pn][pn] when h != NULL, else the gathered code vectors E[idx[b][p]][c] (idx int64 [B][pn*pn], E [V][C]). */
int xq_ms_upsample(const float *h, const float *E, const int64_t *idx, int B, int C, int pn, int H, int W, int bicubic, float *u,
                   xq_stream_t stream);
/* f_hat += Phi(u), Phi(u) = u * (1 - ratio) + (conv3x3(u, phi_w [C][C][3][3]) + phi_b [C]) * ratio (quant.py:261-268);
 * phi_w = phi_b = NULL: Phi = identity.  In place on f_hat [B][C][H][W]. */
int xq_ms_phi_accumulate(const float *u, int B, int C, int H, int W, const float *phi_w, const float *phi_b, float ratio,
                         float *f_hat, xq_stream_t stream);
/* out[B][C][pn][pn] = F.interpolate(in [B][C][H][W], (pn, pn), mode='area') */
int xq_ms_area_pool(const float *in, int B, int C, int H, int W, int pn, float *out, xq_stream_t stream);

/* ---- fp32 forward kernels of the encoder / decoder layers: the reference-parity path (csrc/xq_f32.hip).  Exact fp32 fma chains
 *      on v_mfma_f32_32x32x2_f32, IEEE expf / sqrt / division; inference, plus the three products of nn.Linear's training step.
 *      Activations NHWC fp32. ------------------------------------------------------------------------------------------------------ */
/* w_packed[n][(ky * KW + kx) * Cin + c] = w_oihw[n][c][ky][kx] */
int xq_conv2d_f32_pack_weights(const float *w_oihw, int Cout, int Cin, int KH, int KW, float *w_packed, xq_stream_t stream);
/* y[B][Ho][Wo][Cout] = conv(x[B][Hi][Wi][Cin]) (+ bias): kernel 1x1 or 3x3, stride 1 or 2, pad_top / pad_left zeros before the
 * first row / column, as many zero rows / columns after the last as (Ho, Wo) imply (Conv2d padding p: pad_top = pad_left = p;
 * Downsample, xqgan_model.py:697-704: pads 0 and Ho = Hi / 2); upsample2x = 1: the conv sees the nearest-neighbour 2x upsampling
 * of x (Upsample, :682-686) without it being materialised.  Hi = Wi = Ho = Wo = 1, B = rows: y = x . w^T + b (nn.Linear). */
int xq_conv2d_f32_nhwc(const float *x, const float *w_packed, const float *bias, int B, int Hi, int Wi, int Cin, int Cout, int KH, int KW,
                       int stride, int pad_top, int pad_left, int Ho, int Wo, int upsample2x, float *y, xq_stream_t stream);
/* c[Na][Nb] = a[M][Na]^T . b[M][Nb] in fp32, each output one ascending fma chain over the M rows (deterministic): the weight gradient
 * dW = g^T x of nn.Linear on the fp32 parity path (its forward and data gradient are xq_conv2d_f32_nhwc with Hi = Wi = 1 on W and on
 * W^T).  M = 0 writes zeros. */
int xq_gemm_f32_tn(const float *a, const float *b, int64_t M, int Na, int Nb, float *c, xq_stream_t stream);
/* out[b][i][h][:] = sum_j softmax_j(q_i . k_j * scale) v_j per (batch, head); element (b, token t, head h, d) of q / k / v lives at
 * b * batch_stride + t * token_stride + h * hd + d (ViT packed qkv: three pointers into one [B][N][3][H][hd] buffer;
 * CNN AttnBlock, xqgan_model.py:635-659: H = 1, hd = C); out is [B][N][H][hd] contiguous. */
int xq_attention_f32(const float *q, const float *k, const float *v, int B, int N, int H, int hd, int64_t batch_stride,
                     int64_t token_stride, float scale, float *out, xq_stream_t stream);
/* the same with lse [B][H][N] (nullable) = log sum_j exp(scale q_i . k_j) written beside it, and the backward of it for the fp32 TRAINING leg
 * of the parity tests (head dim <= 64): dq / dk / dv in the layout of q / k / v (same strides: three pointers into one packed gradient
 * buffer), dout [B][N][H][hd] contiguous, delta [B][H][N] workspace (dO_i . O_i).  Deterministic: every output is one ascending chain. */
int xq_attention_f32_lse(const float *q, const float *k, const float *v, int B, int N, int H, int hd, int64_t batch_stride,
                         int64_t token_stride, float scale, float *out, float *lse, xq_stream_t stream);
int xq_attention_f32_backward(const float *q, const float *k, const float *v, const float *out, const float *dout, const float *lse, int B, int N,
                              int H, int hd, int64_t batch_stride, int64_t token_stride, float scale, float *dq, float *dk, float *dv,
                              float *delta, xq_stream_t stream);
/* y = [silu](GroupNorm(x)) on x [B][HW][C] fp32, G groups, statistics accumulated in double (xqgan_model.py:662-672) */
int xq_groupnorm_silu_f32(const float *x, const float *w, const float *bias, int B, int HW, int C, int G, float eps, int silu, float *y,
                          xq_stream_t stream);

/* ---- bf16 GEMMs of the transformer blocks (csrc/xq_gemm.hip; replaces the cuBLAS / hipBLASLt calls behind nn.Linear:
 *      dino_enc/vision_transformer.py:145-197 Attention.qkv / proj, :295-339 Block -> Mlp.fc1 / fc2, :684-692 patch embedding,
 *      dino_enc/to_pixel.py:70-86).  bf16 operands, fp32 accumulation on v_mfma_f32_32x32x16_bf16, one rounding to bf16.
 *      All matrices row-major and contiguous.  impl selects the schedule (tests / benchmarks); use XQ_GEMM_AUTO. ---------- */
#define XQ_GEMM_AUTO 0
#define XQ_GEMM_SIMPLE 1     /* two LDS buffers, one barrier per K tile (any shape the op accepts)                           */
#define XQ_GEMM_RING 2       /* 8-slot LDS-DMA ring, counted vmcnt, staggered wave rows; one workgroup per tile              */
#define XQ_GEMM_PERSISTENT 3 /* the ring kept streaming across a per-CU list of work items; tail tiles / weight gradients
                                cut along K into fp32 slabs (256-column tiles, >= 2 K tiles per item)                       */
#define XQ_GEMM_DUO 4        /* round 6: 128 x 256 tiles, two workgroups per CU (4 waves per SIMD), 5-slot LDS-DMA ring, one workgroup per
                                tile: one workgroup's store burst / prologue runs under the other's K loop (NT / NN, N >= 256)            */
#define XQ_GEMM_PDUO 5       /* round 6: the duo schedule kept streaming across a per-workgroup list of work items (grid = 2 workgroups per CU),
                                fragment reads software-pipelined under the MFMAs, tail tiles cut along K into fp32 slabs                       */
#define XQ_GEMM_WIDE_TILES 0x100 /* OR-ed into impl: 256-column tiles even when N is not a multiple of 256 (ragged last tile) */
#define XQ_GEMM_DEBUG_NO_STORE 0x200 /* OR-ed into impl (NT, persistent schedule): skip the output stores — timing experiments, result unusable */
#define XQ_GEMM_TILE_MAJOR 0x400 /* OR-ed into impl (TN, persistent schedule): execute the K-split items tile-major instead of split-major (A/B timing) */
#define XQ_GEMM_PLAIN_STORE 0x800 /* OR-ed into impl (persistent schedule): plain instead of non-temporal output stores (A/B timing) */
#define XQ_GEMM_TRACE_SUMS 0x40000 /* OR-ed into impl (plain NT / NN / TN, persistent schedule) with a trace buffer bound (xq_gemm_trace_bind): run the
                                      clock-summing twin of the persistent kernel (diagnostics; results unchanged) — per-phase shader-clock differences
                                      summed in scalar registers over the whole kernel; the traced workgroup's 8 waves write [8] = phases summed,
                                      [9] = sum of (phase start -> arrival at the first barrier), [10] = (-> passed), [11] = (-> arrival at the second
                                      barrier = the MFMA segment), [12] = (-> next phase start), [13] = items of the workgroup */
#define XQ_GEMM_OP_NT 0
#define XQ_GEMM_OP_NN 1
#define XQ_GEMM_OP_TN 2
#define XQ_PROF_GEMM 4       /* gemm_*_kernel: 2*M*N*K flops per launch (xq_prof_collect_kind)                               */
/* diagnostics (no reference counterpart): where the clock sums of XQ_GEMM_TRACE_SUMS launches go.  buf = device memory, uint64
 * [8 waves][cap_per_wave >= 16]; the 8 waves of workgroup `workgroup` write entries [8..13] (see XQ_GEMM_TRACE_SUMS).  buf = NULL unbinds.
 * tools/gemm_timeline.py prints the table. */
int xq_gemm_trace_bind(void *buf, int cap_per_wave, int workgroup);
/* bytes of workspace for op (XQ_GEMM_OP_*) at output rows M, columns N, reduction depth K (TN: M = P, N = Q, K = R);
 * NT / NN run without one (workspace NULL: no K-split of the tail tiles). */
size_t xq_gemm_bf16_workspace_bytes(int op, int64_t M, int64_t N, int64_t K);
/* forward: y[M][N] = x[M][K] . w[N][K]^T (+ bias[N], fp32, nullable).  K % 64 == 0, N % 8 == 0, N >= 32. */
int xq_gemm_bf16_nt(const void *x, const void *w, const float *bias, int64_t M, int64_t N, int64_t K, void *y, void *workspace,
                    size_t workspace_bytes, int impl, xq_stream_t stream);
/* data gradient: g_x[M][N] = g_y[M][K] . w[K][N]  (w = the forward's weight [out = K][in = N], read in place with
 * transpose reads: no transposed copy of the weights exists).  K % 64 == 0, N % 8 == 0, N >= 32. */
int xq_gemm_bf16_nn(const void *g_y, const void *w, int64_t M, int64_t N, int64_t K, void *g_x, void *workspace,
                    size_t workspace_bytes, int impl, xq_stream_t stream);
/* fc1 of the transformer MLP with its GELU in the epilogue (timm Mlp.fc1 -> act, vision_transformer.py:295-339):
 * h[M][N] = x . w^T + bias (bf16, kept for the backward), h_act[M][N] = GELU(h) evaluated on the bf16-rounded h exactly as the
 * unfused Linear -> GELU pair does; approximate_tanh selects F.gelu(approximate='tanh').  N >= 256, K >= 128, K % 64 == 0.
 * workspace: xq_gemm_bf16_workspace_bytes(XQ_GEMM_OP_NT, M, N, K) bytes (the tiles beyond the last full round of CUs are cut along K
 * into fp32 slabs whose sum gets the same bias + activation; without a workspace every tile runs whole: same values up to the fp32
 * summation order, a partial last round).  h may be NULL (round 5): a forward that no backward follows (frozen teacher, the discriminator
 * update's pass over the frozen DINO-S trunk) does not write the pre-activation — half the output bytes of this store-bound product. */
int xq_gemm_bf16_nt_gelu(const void *x, const void *w, const float *bias, int64_t M, int64_t N, int64_t K, void *h, void *h_act,
                         int approximate_tanh, void *workspace, size_t workspace_bytes, xq_stream_t stream);
/* data gradient of fc2 with the GELU derivative in the epilogue: g_h[M][N] = (g_y[M][K] . w[K][N]) * GELU'(h[M][N]) (the product
 * rounded to bf16 before the multiplication, as the unfused pair does); colpart (nullable) fp32 [xq_gemm_colpart_rows(M)][N]
 * receives partial column sums of g_h (per 128-row block or per 64-row wave tile) in its first xq_gemm_colpart_rows_written(M, N) rows — the
 * count depends on the schedule in force for the shape; the remaining rows are left untouched: their sum over the written rows is the fc1 bias
 * gradient.  workspace: xq_gemm_bf16_workspace_bytes(XQ_GEMM_OP_NN, M, N, K) bytes, used as in xq_gemm_bf16_nt_gelu.
 * xq_gemm_bf16_nt_gelu_bwd: the same product on the TRANSPOSED weight w_t [N][K] (= W2^T, e.g. the shadow xq_transpose_bf16_batched maintains):
 * both operands K-major, no transpose reads; bit-identical g_h and colpart. */
size_t xq_gemm_colpart_rows(int64_t M);
size_t xq_gemm_colpart_rows_written(int64_t M, int64_t N);
/* schedule of the two fused MLP products (XQ_GEMM_AUTO / _PERSISTENT / _DUO / _PDUO; default: environment XQ_GEMM_FUSED_SCHEDULE, else AUTO);
 * returns the previous value.  Tests and benchmarks; every schedule writes the same h / h_act / g_h bits (K-split tail tiles: up to the fp32
 * summation order) and colpart rows whose column sums agree to the summation order. */
int xq_gemm_fused_schedule(int impl);
int xq_gemm_bf16_nn_gelu_bwd(const void *g_y, const void *w, const void *h, int64_t M, int64_t N, int64_t K, void *g_h, float *colpart,
                             int approximate_tanh, void *workspace, size_t workspace_bytes, xq_stream_t stream);
int xq_gemm_bf16_nt_gelu_bwd(const void *g_y, const void *w_t, const void *h, int64_t M, int64_t N, int64_t K, void *g_h, float *colpart,
                             int approximate_tanh, void *workspace, size_t workspace_bytes, xq_stream_t stream);
/* 3x3 convolution on the GEMM tile engine (implicit GEMM: the A operand is gathered from the NHWC image tap by tap by the
 * LDS-DMA, out-of-image taps read a zero page; csrc/xq_gemm.hip Stager<KMAJOR_CONV>).  x [B][Hi][Wi][Cin] bf16, w_packed
 * [Cout][9 * Cin] bf16 as xq_conv3x3_pack_weights writes it, y [B][Ho][Wo][Cout] bf16 (+ bias, optional ReLU).
 *   forward:     input pixel (oy * stride + ky - pad, ox * stride + kx - pad); upsample2x: of the nearest-2x upsampled image
 *                (Upsample, xqgan_model.py:682-686); stride 2 / pad 0 with Ho = Hi / 2 is Downsample (:697-704);
 *   transposed:  the data gradient of a forward conv of that stride / pad: x = dY [B][Hi][Wi][Cin = forward Cout], w_packed = the
 *                for_data_grad pack, y = dX [B][Ho][Wo] (Ho x Wo = the forward input size).
 * out_mask (nullable): bf16 [B][Ho][Wo][Cout]; y is zeroed where out_mask <= 0 (see xq_conv3x3_nhwc_bf16).
 * Cin % 64 == 0, Cout % 8 == 0, Cout >= 64. */
int xq_conv3x3_gemm_bf16(const void *x, const void *w_packed, const float *bias, int B, int Hi, int Wi, int Cin, int Cout, int Ho, int Wo,
                         int stride, int pad, int upsample2x, int transposed, int relu, const void *out_mask, void *y, int impl,
                         xq_stream_t stream);
/* `batch` independent products in one launch (matrix i at a + i * stride_a etc., strides in elements): op NT / NN write bf16
 * c[M][N], op TN (a [K][M], b [K][N], K % 64 == 0) writes fp32 c[M][N].  Used by the single-head spatial attention of the CNN
 * AttnBlock (xqgan_model.py:646-656: 256 positions x 512 channels per image). */
int xq_gemm_bf16_batched(int op, const void *a, const void *b, int batch, int64_t M, int64_t N, int64_t K, int64_t stride_a,
                         int64_t stride_b, int64_t stride_c, void *c, xq_stream_t stream);
/* weight gradient: g_w[P][Q] (fp32) = g_y[R][P]^T . x[R][Q], the token axis R split over the chip into fp32 slabs in
 * `workspace` that a second kernel sums in a fixed order (deterministic, no atomics).
 * P, Q multiples of 8 and >= 32; any R >= 0. */
int xq_gemm_bf16_tn(const void *g_y, const void *x, int64_t R, int64_t P, int64_t Q, float *g_w, void *workspace,
                    size_t workspace_bytes, int impl, xq_stream_t stream);

/* ---- measurement hooks (bench.py): HIP events recorded around the instrumented hand-written kernels on the stream they
 *      are launched on.  xq_prof_enable(1) resets and arms, (0) disarms and keeps what was recorded, (2) arms without resetting; xq_prof_collect_kind synchronises the recorded events of one
 *      kernel kind and returns their summed duration, launch count and summed algorithmic work (flops) since arming;
 *      xq_prof_collect = the XQ_PROF_ASSIGN kind + reset (kept for older callers). ------------------------------------ */
#define XQ_PROF_ASSIGN 0     /* assign_kernel: 2*N*Vpad*C flops per launch                                        */
#define XQ_PROF_CONV3X3 1    /* conv3x3_kernel: 2*B*H*W*9*Cin*Cout flops per launch                               */
#define XQ_PROF_ATTN_FWD 2   /* attn_fwd_kernel: 4*B*H*N*N*64 flops per launch (2 tile products)                  */
#define XQ_PROF_ATTN_BWD 3   /* dQ (+ delta in its prologue) + dK/dV kernels: 10*B*H*N*N*64 algorithmic flops (5 products; 7 are run) */
/* HBM-bound kernels (round 5; work = ALGORITHMIC BYTES per launch, bench.py prices them against 8 TB/s): */
#define XQ_PROF_RES_LN_FWD 5 /* res_ln_fwd_kernel: rows*D*(4 + y + 4 + a) bytes (x read, branch output read, x_new write, LN output write; y/a 2 B bf16) */
#define XQ_PROF_RES_LN_BWD 6 /* res_ln_bwd_kernel (+ its partials' finalize): rows*D*(a + 4 + 4 + y + 4 + y) bytes (g_a, g_xnew, x_new, y read; g_x, g_y write) */
#define XQ_PROF_ADAMW 7      /* adamw_ema_kernel: n*(20 read + 20 written [+ 2 bf16 shadow]) bytes */
#define XQ_PROF_GROUPNORM 8  /* GroupNorm+SiLU NHWC bf16: forward 3 reads + 1 write (8 B/elem), backward 4 reads + 1 write (10 B/elem) */
#define XQ_PROF_VQ_ELEM 9    /* vq_finish_kernel / vq_backward_kernel + vq_codebook_grad_kernel: 4C B/token per tensor touched + V*C*4 */
#define XQ_PROF_CONV_FROM3 10 /* conv3x3_from3_mfma_kernel: B*H*W*(3*in + Cout*2) bytes */
int xq_prof_enable(int on);
int xq_prof_collect_kind(int kind, double *ms_total, int *launches, double *work_total);
int xq_prof_collect(double *assign_ms_total, int *assign_launches);
/* per-launch (milliseconds, algorithmic work) of the recorded launches of `kind`, in launch order, up to cap entries; returns the
 * number of recorded launches of that kind (may exceed cap), -1 on an event error */
int xq_prof_entries(int kind, double *ms_out, double *work_out, int cap);
/* adds `delta` (may be negative) to the algorithmic work recorded for the most recent launch of `kind`: for launches that run
 * zero-padded operands (conv1_1 data gradient: 3 input channels padded to 64), so that only un-padded flops are reported */
int xq_prof_add_work(int kind, double delta);
/* launches an empty kernel (xq_marker_kernel) with `id` workgroups of 64 threads: a section boundary that shows up in a
 * rocprofv3 kernel trace (tools/rocpd_sections.py attributes the kernels between two markers to a section) */
int xq_prof_marker(int id, xq_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* XQ_OPS_H */
