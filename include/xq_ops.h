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

#define XQ_ABI_VERSION 1

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
 *   C in {8,16,32,64,128,256}; V >= 1.
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
 *   g_z [B][C][HW] is overwritten; g_E [V][C] is ACCUMULATED into (caller zeroes it).
 */
int xq_vq_backward(const float *z, int B, int C, int HW, const float *E, int V, int codebook_norm,
                   const int64_t *idx, const float *g_out, const float *g_vq, const float *g_commit, float beta,
                   float *g_z, float *g_E, xq_stream_t stream);

/* ---- measurement hooks (bench.py): HIP events recorded around the dominant kernel (assign_kernel) on the
 *      stream it is launched on.  xq_prof_enable(1) resets and arms, xq_prof_collect synchronises the
 *      recorded events and returns the summed duration and launch count since arming. ------------------ */
int xq_prof_enable(int on);
int xq_prof_collect(double *assign_ms_total, int *assign_launches);

#ifdef __cplusplus
}
#endif
#endif /* XQ_OPS_H */
