"""M1 / T1 backward: one training backward of VQModel at model level against the REFERENCE's autograd (xqgan_train.py:439-462,
xqgan_model.py:268-365), BASELINE configs 2-5 (ViT-B, P = 1 / 2, single scale / 10-scale ladder, quantizer dropout, latent
perturbation, semantic branch) and, since round 5, config 1 on the CNN encoder / decoder (xqgan_model.py:454-704; the reference's own
Encoder / Decoder / ResnetBlock / AttnBlock / Up / Downsample run UNSHIMMED by oracle/make_golden.py gen_train_cnn; 38 tapped tensors: conv_in /
conv_out of both halves, GroupNorm scales / shifts at five depths, AttnBlock q / k / v / proj_out, the strided and the up-sampling convs, the 1x1
shortcuts, quant_conv / post_quant_conv, the codebook) — the implicit-GEMM data / weight gradient, GroupNorm backward and spatial-attention
backward kernels pinned to the reference's autograd instead of to ATen-GPU op by op.

Goldens (oracle/make_golden.py gen_train_backward, tests/golden/train_bwd_*.npz): the unmodified reference model on the host,
deterministic weights, every random draw recorded; loss = mse(recons, imgs) + vq + commit + entropy + semantic + dependency (the
LPIPS / GAN terms need downloaded checkpoints); gradients of ~25 parameter tensors spread over the whole model (decoder last layer,
1x1 convs around the quantizer, codebooks, Phi convs, first / last transformer blocks, position / token tables), sub-sampled, each
recorded twice — fp32, and under torch.autocast('cpu', bfloat16): the reference's own reduced-precision backward.

Checked here on the MI355X:
  (a) fp32 step (HIP quantizers + perturbation with their hand-written backward; every nn.Linear — forward, data gradient, weight
      gradient — on the hand-written fp32-MFMA kernels of csrc/xq_f32.hip since round 4 (ops_dense.LinearFn's fp32 branch / ops_f32.LinearF32Fn;
      asserted below through nn_ops.IMPL) and the attention on attention_f32_kernel / attention_f32_bwd_q / _kv kernels; LayerNorm / residual /
      GELU run on the fp32 instantiations of the fused row kernels; what is left to ATen are element-wise glue ops): every tapped gradient
      within REL_F32 of the reference's fp32 gradient (relative L2 over the sub-sample);
  (b) the bf16 TRAINING path (hand-written bf16 MFMA GEMMs fwd / dgrad / wgrad, attention fwd / bwd, fused row kernels, quantizer
      backward): its distance to the reference's fp32 gradient is bounded by the distance of the reference's OWN bf16-autocast
      backward to it (x SLACK + FLOOR) — tensor by tensor;
  (b') config 1 (round 6): leg (b) once more with the reference's fp32 code indices teacher-forced on both sides — see SLACK_TF below;
  (c) cfg 2: the parameters after ONE fused AdamW step (xq_adamw_ema_step) equal torch.optim.AdamW applied to the reference's
      gradient (first Adam step: p - lr g / (|g| + eps)), which closes the train step T1 end to end.
A token on an fp32 near-tie may pick another code than the reference's ATen build (the forward test allows 3 % of the pixels for
that); such a flip changes the gradient of a few codebook rows — the bounds below are on whole-tensor relative L2 and absorb it."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle.det_init import det_state_dict
from test_train_forward_parity import CASES, COMMON

pytestmark = pytest.mark.gpu

REL_F32 = 1e-4          # (a): fp32 path vs reference fp32 (measured: <= 1.1e-5 on cfg 2 / 3 / 4, profiles/r03_gradient_parity.txt)
# cfg 5 (RobustTok): add_perturbation picks, per token, the code of RANK r among the 100 nearest (latent_perturbation.py:20-24) —
# neighbouring ranks are separated by ~1e-6 in distance, so the fp32 rounding of the encoder (ATen's GPU GEMMs in this fp32 TRAINING
# pass vs ATen's CPU GEMMs in the reference) moves a few of the 128 randomly ranked tokens to the neighbouring code
# ("parity on exact ties is only defined up to the chosen code's distance").  The codebook gradient (vq loss only: no
# perturbed sample reaches it) still matches to 1e-6; everything downstream of the perturbed sample moves by <= ~1.7e-2.
REL_F32_PERTURBED = 2.5e-2
SLACK, FLOOR = 1.5, 3e-3   # (b): err_ours <= SLACK * err_reference_bf16 + FLOOR
# (b') config 1, round 6: the same bound against the reference's bf16 backward with the fp32 leg's code indices TEACHER-FORCED (golden keys
# "bf16tf:*", oracle/make_golden.py gen_train_cnn).  At B = 4 the reference's own bf16 gradient error is 6 - 38 % per tensor (median 8.7 %) because
# 35 of its 1024 tokens pick another code under bf16 — a bound a 10 %-wrong data / weight-gradient kernel passes.  With the indices pinned on
# both sides the reference's bf16 error is 0.3 - 2.8 % (median 1.0 %): the level the ViT configs are held to, and what pins the hand-written
# implicit-GEMM dgrad / wgrad, GroupNorm backward and spatial-attention backward kernels (xqgan_model.py:454-704) to the reference's autograd.
SLACK_TF, FLOOR_TF = 1.5, 3e-3


def _force_indices(monkeypatch, idx_np):
    """ops.vq_forward_raw with the picks replaced by idx_np: the kernel runs (its own picks are counted), then z_q / histogram / squared error are
    rebuilt from the forced picks by the reference's formulas (xqgan_model.py:753-799) in fp32 ATen — test infrastructure, the product is untouched;
    the hand-written backward (xq_vq_backward) then scatters through the forced picks it finds in the autograd context."""
    import torch.nn.functional as F
    from imagefolder_amd import ops
    real = ops.vq_forward_raw
    forced = torch.from_numpy(np.asarray(idx_np).astype(np.int64).reshape(-1))
    state = {"flips": None}

    def patched(z, codebook, codebook_norm, ste, want_zq=True, want_hist=False, want_loss=False):
        zq, idx, hist, loss = real(z, codebook, codebook_norm, ste, want_zq, want_hist, want_loss)
        if idx.numel() != forced.numel():
            return zq, idx, hist, loss
        fi = forced.to(idx.device)
        state["flips"] = int((idx != fi).sum())
        zf = z.detach().float()
        B, C = zf.shape[0], zf.shape[1]
        zt = zf.reshape(B, C, -1).permute(0, 2, 1).reshape(-1, C)
        E = codebook.detach().float()
        if codebook_norm:
            zt, E = F.normalize(zt, p=2, dim=-1), F.normalize(E, p=2, dim=-1)
        q = E[fi]
        if codebook_norm:
            q = F.normalize(q, p=2, dim=-1)
        tok = zt + (q - zt) if ste else q
        zq2 = tok.reshape(B, -1, C).permute(0, 2, 1).reshape(zf.shape).contiguous()
        return (zq2 if want_zq else None, fi, torch.bincount(fi, minlength=E.shape[0]).float() if want_hist else None,
                ((q - zt) ** 2).sum().reshape(1) if want_loss else None)
    monkeypatch.setattr(ops, "vq_forward_raw", patched)
    return state


def _sub(t):
    f = t.reshape(-1)
    k = max(1, (f.numel() + 16383) // 16384)
    return f[::k]


def _rel(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b.astype(np.float64)) / max(np.linalg.norm(b.astype(np.float64)), 1e-30))


def _run(name, amp_dtype, monkeypatch, lr=1e-4, max_grad_norm=0.0):
    """one TokenizerTrainStep.step of the mirror with the reference's draws replayed; returns ({param: grad}, {param: value after the step})"""
    from imagefolder_amd import latent_perturbation, xqgan_model
    from imagefolder_amd.dino_enc.vision_transformer import DropPath
    from imagefolder_amd.train import TokenizerTrainStep
    fwd_name = name.replace("train_bwd_", "train_fwd_")
    g = load_golden(fwd_name)
    seed, B = int(g["seed"]), int(g["B"])
    torch.manual_seed(seed)
    m = xqgan_model.VQ_models["VQ-16"](**dict(COMMON, **CASES[fwd_name])).train()
    m.load_state_dict(det_state_dict(m.state_dict(), seed))
    m = m.cuda()
    x = (torch.rand(B, 3, 256, 256, generator=torch.Generator().manual_seed(4321 + seed)) * 2 - 1).cuda()
    DropPath.REPLAY = [torch.from_numpy(r) for r in g["droppath"]]
    real_randint = torch.randint

    def randint(*a, **k):
        if len(g["dropout_rand"]) and len(a) >= 3 and tuple(a[2]) == (B,):
            return torch.from_numpy(g["dropout_rand"]).clone()
        return real_randint(*a, **k)
    monkeypatch.setattr(torch, "randint", randint)
    if len(g["lp_prob"]):
        alpha = float(g["alpha"])
        prob, ridx = torch.from_numpy(g["lp_prob"]), torch.from_numpy(g["lp_idx"])
        rank = torch.where(prob > alpha, torch.zeros_like(ridx), ridx)

        def draw(n_tokens, a_, delta_, device):
            return rank.to(device)
        monkeypatch.setattr(latent_perturbation, "draw_ranks", draw)

    def gen_loss(out, imgs):
        recons, (vq, commit, entropy, usages), sem, detail, dep = out
        return torch.nn.functional.mse_loss(recons.float(), imgs) + vq + commit + entropy + (0.0 if sem is None else sem) + dep
    ts = TokenizerTrainStep(m, gen_loss, lr=lr, betas=(0.9, 0.95), weight_decay=0.0, eps=1e-8, use_ema=False, amp_dtype=amp_dtype,
                            max_grad_norm=max_grad_norm)
    names = [n for n, p in m.named_parameters() if p.requires_grad]
    grads = {}
    opt_step = ts.opt.step

    def recording_step():
        for n, p, o in zip(names, ts.arena.params, ts.arena.offsets):
            grads[n] = ts.arena.g[o:o + p.numel()].clone()
        opt_step()
    ts.opt.step = recording_step
    try:
        loss = ts.step(x, 0, float(g["alpha"]), float(g["beta"]), int(g["delta"]))
        torch.cuda.synchronize()
        assert DropPath.REPLAY == [], f"{len(DropPath.REPLAY)} recorded DropPath masks were not consumed"
    finally:
        DropPath.REPLAY = None
    after = {n: p.detach().clone() for n, p in m.named_parameters() if p.requires_grad}
    if max_grad_norm:
        after["__grad_norm__"] = float(ts.opt.last_grad_norm)
    return float(loss), grads, after, seed


@pytest.mark.parametrize("name", ["train_bwd_cfg1_cnn_vq4096", "train_bwd_cfg2_vq8192", "train_bwd_cfg3_vp2_16384", "train_bwd_cfg4_msvr10p2_4096",
                                  "train_bwd_cfg5_robusttok"])
def test_model_level_gradients_match_the_reference_autograd(name, monkeypatch):
    gb = load_golden(name)
    taps = [str(t) for t in gb["taps"]]
    assert len(taps) >= 20
    cnn = name == "train_bwd_cfg1_cnn_vq4096"
    # ---- (a) fp32 ----
    lr = 1e-4
    from imagefolder_amd import nn_ops
    monkeypatch.setattr(nn_ops, "F32_TRAIN_LINEAR", True)      # the parity kernels of csrc/xq_f32.hip (off by default: 1/16 of the bf16 rate)
    for key in ("linear_fp32_training", "linear_library", "attention_fp32_training", "attention_library", "attention", "conv2d_library",
                "group_norm_library", "conv2d", "group_norm_silu", "spatial_attention", "conv2d_downsample", "conv2d_upsample"):
        nn_ops.IMPL.pop(key, None)
    # config 1 also runs the clipping pass (max_grad_norm far above the norm: coefficient 1) to read the GLOBAL gradient norm the trainer's
    # clip_grad_norm_ sees (xqgan_train.py:456-458) against the reference's
    loss32, g32, after32, seed = _run(name, None, monkeypatch, lr=lr, max_grad_norm=1e30 if cnn else 0.0)
    if cnn:
        np.testing.assert_allclose(after32["__grad_norm__"], float(gb["gnorm_f32"]), rtol=1e-4)
    if not cnn:
        # the Linear layers (forward, data gradient, weight gradient) and the attention (forward, backward) of this leg ran on the hand-written
        # fp32 kernels, none on the library
        assert nn_ops.IMPL.get("linear_fp32_training", "").startswith("hip") and "linear_library" not in nn_ops.IMPL
        assert nn_ops.IMPL.get("attention_fp32_training", "").startswith("hip") and "attention_library" not in nn_ops.IMPL
        assert not nn_ops.IMPL.get("attention", "").startswith("library")
    # (config 1: the fp32 TRAINING step of the CNN runs the convolutions / GroupNorm on ATen-GPU — this leg pins the module wiring, the
    #  quantizer's hand-written backward and the 1x1 convs around it to the reference's autograd; the hand-written implicit-GEMM forward /
    #  data-gradient / weight-gradient, GroupNorm and spatial-attention kernels are what leg (b) runs, asserted there)
    np.testing.assert_allclose(loss32, float(gb["loss_f32"]), rtol=1e-4 if name == "train_bwd_cfg5_robusttok" else 5e-6)
    rows = []
    for n in taps:
        ref = gb[f"f32:{n}"]
        got = _sub(g32[n]).float().cpu().numpy()
        rows.append((n, _rel(got, ref), float(gb[f"f32:{n}:l2"])))
    # ---- (b) bf16 training kernels ----
    monkeypatch.setattr(nn_ops, "STRICT_HIP", True)      # a dense op of the bf16 step that drops to a library raises
    loss16, g16, _, _ = _run(name, torch.bfloat16, monkeypatch, lr=lr)
    monkeypatch.setattr(nn_ops, "STRICT_HIP", False)
    if cnn:      # every dense op of the bf16 step of the CNN ran on a hand-written kernel
        for key in ("conv2d", "group_norm_silu", "spatial_attention", "conv2d_downsample", "conv2d_upsample", "conv2d_from_rgb", "conv2d_to_rgb"):
            assert nn_ops.IMPL.get(key, "").startswith("hip"), (key, nn_ops.IMPL.get(key))
    rows16 = []
    for n in taps:
        ref32, ref16 = gb[f"f32:{n}"], gb[f"bf16:{n}"]
        got = _sub(g16[n]).float().cpu().numpy()
        rows16.append((n, _rel(got, ref32), _rel(ref16, ref32)))
    rows_tf = []
    if cnn:      # (b') the bf16 training kernels with the reference's fp32 code indices forced, against the reference's bf16 leg forced the same way
        st = _force_indices(monkeypatch, load_golden(name.replace("train_bwd_", "train_fwd_"))["idx"])
        monkeypatch.setattr(nn_ops, "STRICT_HIP", True)
        loss_tf, g_tf, _, _ = _run(name, torch.bfloat16, monkeypatch, lr=lr)
        monkeypatch.setattr(nn_ops, "STRICT_HIP", False)
        monkeypatch.undo()      # (restores vq_forward_raw and everything _run patched; the legs above are complete)
        assert st["flips"] is not None, "the forced-index hook never saw the training forward"
        for n in taps:
            ref32, reftf = gb[f"f32:{n}"], gb[f"bf16tf:{n}"]
            rows_tf.append((n, _rel(_sub(g_tf[n]).float().cpu().numpy(), ref32), _rel(reftf, ref32)))
        print(f"\n{name} teacher-forced bf16 leg: loss {loss_tf:.6f} (reference bf16, forced: {float(gb['loss_bf16tf']):.6f}; fp32 {float(gb['loss_f32']):.6f}); "
              f"our bf16 encoder had moved {st['flips']} of 1024 tokens, the reference's {int(gb['bf16tf_flips_replaced'])}")
        for n, e, r in rows_tf:
            print(f"  {n:48s} bf16 rel {e:8.2e}  (reference's own bf16, indices forced: {r:8.2e})")
    print(f"\n{name}: loss fp32 {loss32:.6f} (ref {float(gb['loss_f32']):.6f}), bf16 {loss16:.6f} (ref bf16 {float(gb['loss_bf16']):.6f})")
    for (n, e32, l2), (_, e16, r16) in zip(rows, rows16):
        print(f"  {n:48s} |g| {l2:9.3e}  fp32 rel {e32:8.2e}   bf16 rel {e16:8.2e}  (reference's own bf16: {r16:8.2e})")
    perturbed = name == "train_bwd_cfg5_robusttok"
    bad32 = [(n, e) for n, e, _ in rows if e > (REL_F32_PERTURBED if perturbed else REL_F32)]
    assert not bad32, f"fp32 gradients off the reference: {bad32}"
    if perturbed:
        assert dict((n, e) for n, e, _ in rows)["quantize.embedding.weight"] <= REL_F32
    bad16 = [(n, e, r) for n, e, r in rows16 if e > SLACK * r + FLOOR]
    assert not bad16, f"bf16 gradients further from the reference's fp32 gradient than its own bf16 backward allows: {bad16}"
    bad_tf = [(n, e, r) for n, e, r in rows_tf if e > SLACK_TF * r + FLOOR_TF]
    assert not bad_tf, f"config 1, indices teacher-forced: bf16 gradients of the hand-written CNN kernels off the reference's forced bf16 backward: {bad_tf}"
    if cnn:
        assert max(r for _, _, r in rows_tf) <= 3e-2 and abs(loss_tf - float(gb["loss_f32"])) <= SLACK_TF * abs(float(gb["loss_bf16tf"]) - float(gb["loss_f32"])) + 2e-3 * abs(float(gb["loss_f32"]))
    assert abs(loss16 - float(gb["loss_f32"])) <= SLACK * abs(float(gb["loss_bf16"]) - float(gb["loss_f32"])) + 2e-3 * abs(float(gb["loss_f32"]))
    # ---- (c) one AdamW step on the reference gradient (cfg 2 closes T1) ----
    if name == "train_bwd_cfg2_vq8192":
        from imagefolder_amd import xqgan_model
        torch.manual_seed(seed)
        p0 = det_state_dict(xqgan_model.VQ_models["VQ-16"](**dict(COMMON, **CASES["train_fwd_cfg2_vq8192"])).state_dict(), seed)
        checked = 0
        for n in taps:
            ref = gb[f"f32:{n}"].astype(np.float64)
            start = _sub(p0[n]).double().numpy()
            want = start - lr * ref / (np.abs(ref) + 1e-8)             # first AdamW step, weight_decay 0: m_hat = g, v_hat = g^2
            got = _sub(after32[n]).double().cpu().numpy()
            solid = np.abs(ref) > 1e-3 * np.abs(ref).max()             # entries whose sign is not rounding noise
            if solid.sum() < 16:
                continue
            ok = np.abs(got - want)[solid] <= 0.02 * lr
            assert ok.mean() >= 0.995, (n, float(ok.mean()))
            checked += 1
        assert checked >= 15
