"""TEST INFRASTRUCTURE — generates tests/golden/*.npz by running the UNMODIFIED reference modules
(imported from /root/reference through oracle/ref_import.py) on CPU in fp32.

The reference ships no tests / golden vectors for this path, so these fixtures are
the pin: inputs + the reference's own outputs (indices, z_q, losses, usage, autograd grads).
Run here (build container) only:   python oracle/make_golden.py
The GPU box never needs /root/reference: tests read the committed .npz files.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.ref_import import load_reference  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def meta():
    return dict(torch_version=torch.__version__, generator="oracle/make_golden.py",
                reference="lxa9867/ImageFolder @ /root/reference (2025-04-18 snapshot)")


def gen_vq(name, V, C, B, H, W, seed, codebook_norm=True, beta=0.25, z_scale=1.0, clustered=False):
    """VectorQuantizer.forward/backward + f_to_idxBl_or_fhat (xqgan_model.py:722-833)."""
    R = load_reference()
    torch.manual_seed(seed)
    q = R["VectorQuantizer"](V, C, beta, codebook_norm).train()
    z = torch.randn(B, C, H, W) * z_scale
    if clustered:
        # realistic failure mode: many tokens compete for few codes (low usage at init)
        centers = q.embedding.weight.detach()[torch.randint(0, V, (8,))]
        pick = torch.randint(0, 8, (B, H, W))
        z = centers[pick].permute(0, 3, 1, 2) * 3.0 + 0.05 * torch.randn(B, C, H, W)
    z.requires_grad_(True)
    zq, usage, vq, commit, _ = q(z)
    g_out = torch.randn_like(zq) * 0.1
    g_vq, g_commit = 1.7, 0.6
    (zq * g_out).sum().add(vq * g_vq).add(commit * g_commit).backward()
    with torch.no_grad():
        idx = q.f_to_idxBl_or_fhat(z.detach(), to_fhat=False, v_patch_nums=None)[0]
        fhat = q.f_to_idxBl_or_fhat(z.detach(), to_fhat=True, v_patch_nums=None)[0]
    np.savez(os.path.join(OUT, name + ".npz"),
             z=z.detach().numpy(), E=q.embedding.weight.detach().numpy(), beta=np.float32(beta),
             codebook_norm=np.int32(codebook_norm), idx=idx.numpy(), zq=zq.detach().numpy(), fhat=fhat.numpy(),
             vq_loss=np.float32(vq.item()), commit_loss=np.float32(commit.item()), usage=np.float32(usage[0]),
             ema_hit=q.ema_vocab_hit_SV.numpy(), g_out=g_out.numpy(), g_vq=np.float32(g_vq),
             g_commit=np.float32(g_commit), g_z=z.grad.numpy(), g_E=q.embedding.weight.grad.numpy(),
             meta=np.array(str(meta())))
    print("wrote", name, "usage", usage[0], "vq", vq.item())


def gen_perturb(name, V, C, B, H, W, seed, alpha, beta, delta, codebook_norm=True):
    """add_perturbation (latent_perturbation.py:4-35), with the RNG draws recorded."""
    R = load_reference()
    torch.manual_seed(seed)
    q = R["VectorQuantizer"](V, C, 0.25, codebook_norm).train()
    z = (torch.randn(B, C, H, W) * (1.0 if codebook_norm else 0.02)).requires_grad_(True)
    with torch.no_grad():
        zq0 = q.f_to_idxBl_or_fhat(z.detach(), to_fhat=True, v_patch_nums=None)[0]
    zq_in = (zq0 + 0.01 * torch.randn_like(zq0)).requires_grad_(True)  # any tensor may arrive as z_q
    N = B * H * W
    torch.manual_seed(seed + 1000)
    out = R["add_perturbation"](z, zq_in, C, codebook_norm, q.embedding, alpha, beta, delta)
    torch.manual_seed(seed + 1000)  # replay the two draws (latent_perturbation.py:21-22)
    random_prob = torch.rand(N)
    random_idx = torch.randint(0, delta, (N,))
    rank = torch.where(random_prob > alpha, torch.zeros_like(random_idx), random_idx)
    g_out = torch.randn_like(out) * 0.1
    (out * g_out).sum().backward()
    np.savez(os.path.join(OUT, name + ".npz"), z=z.detach().numpy(), zq_in=zq_in.detach().numpy(),
             E=q.embedding.weight.detach().numpy(), codebook_norm=np.int32(codebook_norm), alpha=np.float32(alpha),
             beta=np.float32(beta), delta=np.int32(delta), rank=rank.numpy().astype(np.int32), out=out.detach().numpy(),
             g_out=g_out.numpy(), g_z=z.grad.numpy(), g_zq=zq_in.grad.numpy(), meta=np.array(str(meta())))
    print("wrote", name, "n_pert", int(B * beta), "ranks>0:", int((rank[: int(B * beta) * H * W] > 0).sum()))


def gen_msvq(name, V, C, B, pns, seed, codebook_drop=0.1, start_drop=3, using_znorm=True, share=4, var_variant=False):
    """VectorQuantizer2.forward/backward + f_to_idxBl_or_fhat (tokenizer_image/quant.py:64-223) or, with
    var_variant, the original models/quant.py quantizer."""
    R = load_reference()
    torch.manual_seed(seed)
    H = W = pns[-1]
    if var_variant:
        q = R["var_quant"].VectorQuantizer2(V, C, using_znorm, beta=0.25, v_patch_nums=tuple(pns), share_quant_resi=share).train()
    else:
        q = R["VectorQuantizer2"](V, C, using_znorm=using_znorm, v_patch_nums=list(pns), num_latent_tokens=H * W,
                                  share_quant_resi=share, codebook_drop=codebook_drop).train()
    if not using_znorm:
        q.embedding.weight.data.mul_(30.0)  # raw-L2 path: codebook and latents on comparable scales
    f = (torch.randn(B, C, H, W) * 0.6).requires_grad_(True)
    dropout = torch.randint(start_drop, len(pns) + 1, (B,))
    if var_variant:
        f_hat, usages, vq = q(f, ret_usages=True)
        commit = torch.zeros(())
    else:
        f_hat, usages, vq, commit, _ = q(f, ret_usages=True, dropout=dropout)
    g_out = torch.randn_like(f_hat) * 0.05
    g_vq, g_commit = 1.3, 0.7
    ((f_hat * g_out).sum() + vq * g_vq + commit * g_commit).backward()
    with torch.no_grad():
        idx_list = q.f_to_idxBl_or_fhat(f.detach(), to_fhat=False, v_patch_nums=None)
        fhat_list = q.f_to_idxBl_or_fhat(f.detach(), to_fhat=True, v_patch_nums=None)
    convs = list(q.quant_resi.qresi_ls) if share > 1 else [q.quant_resi.qresi]
    SN = len(pns)
    phi_sel = [int(np.argmin(np.abs(q.quant_resi.ticks - si / (SN - 1)))) for si in range(SN)] if share > 1 else [0] * SN
    np.savez(os.path.join(OUT, name + ".npz"), f=f.detach().numpy(), E=q.embedding.weight.detach().numpy(),
             pns=np.array(pns, np.int32), using_znorm=np.int32(using_znorm), codebook_drop=np.float32(codebook_drop),
             dropout=dropout.numpy().astype(np.int32), var_variant=np.int32(var_variant),
             phi_w=np.stack([c.weight.detach().numpy() for c in convs]), phi_b=np.stack([c.bias.detach().numpy() for c in convs]),
             phi_sel=np.array(phi_sel, np.int32), f_hat=f_hat.detach().numpy(), vq_loss=np.float32(vq.item()),
             commit_loss=np.float32(commit.item()), usages=np.array(usages, np.float32), ema_hit=q.ema_vocab_hit_SV.numpy(),
             idx=np.concatenate([i.reshape(-1).numpy() for i in idx_list]), fhat_last=fhat_list[-1].numpy(),
             fhat_scale3=fhat_list[min(3, SN - 1)].numpy(), g_out=g_out.numpy(), g_vq=np.float32(g_vq),
             g_commit=np.float32(g_commit), g_f=f.grad.numpy(), g_E=q.embedding.weight.grad.numpy(),
             g_phi_w=np.stack([c.weight.grad.numpy() for c in convs]), g_phi_b=np.stack([c.bias.grad.numpy() for c in convs]),
             meta=np.array(str(meta())))
    print("wrote", name, "vq", vq.item(), "commit", float(commit), "usages", [round(u, 2) for u in usages][:4])


def gen_var_helpers(name, V, C, B, pns, seed, share=4, var_variant=False):
    """VAR-side helpers of VectorQuantizer2 (tokenizer_image/quant.py:148-180 embed_to_fhat, :226-245 idxBl_to_var_input,
    :248-258 get_next_autoregressive_input; var_variant: the same methods of models/quant.py) on a ladder of ground-truth
    indices produced by the reference's own f_to_idxBl_or_fhat."""
    R = load_reference()
    torch.manual_seed(seed)
    H = W = pns[-1]
    SN = len(pns)
    if var_variant:
        q = R["var_quant"].VectorQuantizer2(V, C, True, beta=0.25, v_patch_nums=tuple(pns), share_quant_resi=share).eval()
    else:
        q = R["VectorQuantizer2"](V, C, using_znorm=True, v_patch_nums=list(pns), num_latent_tokens=H * W,
                                  share_quant_resi=share, codebook_drop=0.0).eval()
    for c in (list(q.quant_resi.qresi_ls) if share > 1 else [q.quant_resi.qresi]):   # non-trivial Phi convs
        torch.nn.init.normal_(c.weight, std=0.2)
        torch.nn.init.normal_(c.bias, std=0.1)
    q.embedding.weight.data.mul_(1.0 + torch.rand(V, 1))      # rows of different norms (lookups are not normalised)
    f = torch.randn(B, C, H, W) * 0.6
    with torch.no_grad():
        idx_list = q.f_to_idxBl_or_fhat(f, to_fhat=False, v_patch_nums=None)
        ms_h = [q.embedding(idx).transpose(1, 2).reshape(B, C, pn, pn).contiguous() for idx, pn in zip(idx_list, pns)]
        fhats = q.embed_to_fhat(ms_h, all_to_max_scale=True, last_one=False)
        fhat_last = q.embed_to_fhat(ms_h, all_to_max_scale=True, last_one=True)
        var_in = q.idxBl_to_var_input(idx_list)
        f_hat = torch.zeros(B, C, H, W)
        nexts = []
        for si in range(SN):
            f_hat, nxt = q.get_next_autoregressive_input(si, SN, f_hat, ms_h[si])
            nexts.append(nxt.clone())
    convs = list(q.quant_resi.qresi_ls) if share > 1 else [q.quant_resi.qresi]
    phi_sel = [int(np.argmin(np.abs(q.quant_resi.ticks - si / (SN - 1)))) for si in range(SN)] if share > 1 else [0] * SN
    np.savez(os.path.join(OUT, name + ".npz"), E=q.embedding.weight.detach().numpy(), pns=np.array(pns, np.int32),
             share=np.int32(share), var_variant=np.int32(var_variant),
             phi_w=np.stack([c.weight.detach().numpy() for c in convs]), phi_b=np.stack([c.bias.detach().numpy() for c in convs]),
             phi_sel=np.array(phi_sel, np.int32), phi_ratio=np.float32(abs(q.quant_resi_ratio)),
             idx=np.concatenate([i.reshape(-1).numpy() for i in idx_list]),
             fhat_scales=np.stack([t.numpy() for t in fhats]), fhat_last=fhat_last.numpy(), var_input=var_in.numpy(),
             next_maps=np.concatenate([t.reshape(-1).numpy() for t in nexts]), f_hat_final=f_hat.numpy(), meta=np.array(str(meta())))
    print("wrote", name, "var_input", tuple(var_in.shape), "|f_hat|", float(f_hat.abs().mean()))


def gen_lfq(name, Cbits, B, pns, seed, codebook_drop=0.1, start_drop=3, using_znorm=True, share=4, entropy_weight=0.1):
    """LFQ.forward/backward (lookup_free_quantize.py:149-250) in train mode with quantizer dropout + f_to_idxBl_or_fhat."""
    R = load_reference()
    torch.manual_seed(seed)
    H = W = pns[-1]
    V = 2 ** Cbits
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):   # the constructor prints the scaler
        q = R["LFQ"](V, Cbits, using_znorm=using_znorm, v_patch_nums=list(pns), num_latent_tokens=H * W, share_quant_resi=share,
                     codebook_drop=codebook_drop, scale=1.0, entropy_weight=entropy_weight, soft_entropy=True).train()
    f = (torch.randn(B, Cbits, H, W) * 0.6).requires_grad_(True)
    dropout = torch.randint(start_drop, len(pns) + 1, (B,))
    f_hat, usages, vq, commit, ent = q(f, ret_usages=True, dropout=dropout)
    g_out = torch.randn_like(f_hat) * 0.05
    g_vq, g_commit, g_ent = 1.3, 0.7, 0.9
    ((f_hat * g_out).sum() + vq * g_vq + commit * g_commit + ent * g_ent).backward()
    with torch.no_grad():
        idx_list = q.f_to_idxBl_or_fhat(f.detach(), to_fhat=False, v_patch_nums=None)
        fhat_list = q.f_to_idxBl_or_fhat(f.detach(), to_fhat=True, v_patch_nums=None)
    convs = list(q.quant_resi.qresi_ls)
    np.savez(os.path.join(OUT, name + ".npz"), f=f.detach().numpy(), Cbits=np.int32(Cbits), pns=np.array(pns, np.int32),
             using_znorm=np.int32(using_znorm), codebook_drop=np.float32(codebook_drop), share=np.int32(share),
             entropy_weight=np.float32(entropy_weight), dropout=dropout.numpy().astype(np.int32),
             phi_w=np.stack([c.weight.detach().numpy() for c in convs]), phi_b=np.stack([c.bias.detach().numpy() for c in convs]),
             f_hat=f_hat.detach().numpy(), vq_loss=np.float32(vq.item()), commit_loss=np.float32(commit.item()),
             entropy_loss=np.float32(ent.item()), usages=np.array(usages, np.float32), ema_hit=q.ema_vocab_hit_SV.numpy(),
             idx=np.concatenate([i.reshape(-1).numpy() for i in idx_list]), fhat_last=fhat_list[-1].numpy(),
             g_out=g_out.numpy(), g_vq=np.float32(g_vq), g_commit=np.float32(g_commit), g_ent=np.float32(g_ent),
             g_f=f.grad.numpy(), g_phi_w=np.stack([c.weight.grad.numpy() for c in convs]),
             g_phi_b=np.stack([c.bias.grad.numpy() for c in convs]), meta=np.array(str(meta())))
    print("wrote", name, "vq", vq.item(), "commit", float(commit), "entropy", float(ent), "usages", [round(u, 2) for u in usages][:4])


def gen_lfq_var_helpers(name, Cbits, B, pns, seed, using_znorm=True, share=4):
    """VAR-side helpers of LFQ (lookup_free_quantize.py:311-343 embed_to_fhat, :404-415 get_next_autoregressive_input) on the
    sign codes of a random latent.  (idxBl_to_var_input :383-401 dereferences `self.embedding`, which LFQ does not define:
    it raises AttributeError upstream, nothing to record.)"""
    R = load_reference()
    torch.manual_seed(seed)
    H = W = pns[-1]
    SN = len(pns)
    import contextlib, io
    with contextlib.redirect_stdout(io.StringIO()):
        q = R["LFQ"](2 ** Cbits, Cbits, using_znorm=using_znorm, v_patch_nums=list(pns), num_latent_tokens=H * W, share_quant_resi=share,
                     codebook_drop=0.0, scale=1.0, entropy_weight=0.0, soft_entropy=True).eval()
    for c in list(q.quant_resi.qresi_ls):
        torch.nn.init.normal_(c.weight, std=0.2)
        torch.nn.init.normal_(c.bias, std=0.1)
    f = torch.randn(B, Cbits, H, W) * 0.6
    with torch.no_grad():
        idx_list = q.f_to_idxBl_or_fhat(f, to_fhat=False, v_patch_nums=None)
        ms_h = [q.indices_to_bits(idx, si).transpose(1, 2).reshape(B, Cbits, pn, pn).float().contiguous()
                for si, (idx, pn) in enumerate(zip(idx_list, pns))]
        fhats = q.embed_to_fhat(ms_h, all_to_max_scale=True, last_one=False)
        fhat_last = q.embed_to_fhat(ms_h, all_to_max_scale=True, last_one=True)
        f_hat = torch.zeros(B, Cbits, H, W)
        nexts = []
        for si in range(SN):
            f_hat, nxt = q.get_next_autoregressive_input(si, SN, f_hat, ms_h[si])
            nexts.append(nxt.clone())
    convs = list(q.quant_resi.qresi_ls)
    phi_sel = [int(np.argmin(np.abs(q.quant_resi.ticks - si / (SN - 1)))) for si in range(SN)]
    np.savez(os.path.join(OUT, name + ".npz"), Cbits=np.int32(Cbits), pns=np.array(pns, np.int32), share=np.int32(share),
             using_znorm=np.int32(using_znorm),
             phi_w=np.stack([c.weight.detach().numpy() for c in convs]), phi_b=np.stack([c.bias.detach().numpy() for c in convs]),
             phi_sel=np.array(phi_sel, np.int32), phi_ratio=np.float32(abs(q.quant_resi_ratio)),
             ms_h=np.concatenate([t.reshape(-1).numpy() for t in ms_h]),
             fhat_scales=np.stack([t.numpy() for t in fhats]), fhat_last=fhat_last.numpy(),
             next_maps=np.concatenate([t.reshape(-1).numpy() for t in nexts]), f_hat_final=f_hat.numpy(), meta=np.array(str(meta())))
    print("wrote", name, "|f_hat|", float(f_hat.abs().mean()))


def gen_model(name, kw, seed):
    """VQModel.img_to_reconstructed_img + code indices (xqgan_model.py:367-403) with deterministic weights
    (oracle/det_init.py); eval mode (no DropPath), fp32 CPU = the reference CPU path of BASELINE config 1/2."""
    from oracle.det_init import det_state_dict
    R = load_reference()
    torch.manual_seed(seed)
    m = R["VQ_models"]["VQ-16"](**kw).eval()
    m.load_state_dict(det_state_dict(m.state_dict(), seed))
    x = torch.rand(1, 3, 256, 256, generator=torch.Generator().manual_seed(1234 + seed)) * 2 - 1
    with torch.no_grad():
        rec = m.img_to_reconstructed_img(x)
        h = m.encoder(x)
        if kw["enc_type"] == "dinov2":
            b, l, c = h.shape
            h = h.view(b, int(l ** 0.5), int(l ** 0.5), c).permute(0, 3, 1, 2)
        f = m.quant_conv(h)
        idx = m.quantize.f_to_idxBl_or_fhat(f, to_fhat=False, v_patch_nums=None)[0]
    np.savez(os.path.join(OUT, name + ".npz"), x=x.numpy(), rec=rec.numpy(), idx=idx.numpy(), f=f.numpy(), seed=np.int32(seed),
             meta=np.array(str(meta())))
    print("wrote", name, "rec range", float(rec.min()), float(rec.max()), "distinct codes", len(set(idx.tolist())))


def gen_tokens(name, kw, B, seed):
    """Code indices of a product-quantizer / multi-scale tokenizer as the REFERENCE emits them at inference
    (xqgan_model.py:386-394: f.chunk(P, dim=2) -> quantizes[i].f_to_idxBl_or_fhat(f_i, to_fhat=False, v_patch_nums)): per branch
    the latent f_i and the list of per-scale index maps, plus the flat token rows in the layout of imagefolder_amd.tokenize
    (branch p: its scales in order, ids offset by p * codebook_size — the shared vocabulary of P * V ids, configs/VP2-16384.yaml)."""
    from oracle.det_init import det_state_dict
    R = load_reference()
    torch.manual_seed(seed)
    m = R["VQ_models"]["VQ-16"](**kw).eval()
    m.load_state_dict(det_state_dict(m.state_dict(), seed))
    x = torch.rand(B, 3, 256, 256, generator=torch.Generator().manual_seed(1234 + seed)) * 2 - 1
    P, V = kw["product_quant"], kw["codebook_size"]
    multi = len(kw["v_patch_nums"]) > 1
    out = {}
    with torch.no_grad():
        h = m.encoder(x)
        b, l, c = h.shape
        h = h.view(b, l, 1, c).permute(0, 3, 1, 2)                       # xqgan_model.py:246-249 (product_quant > 1)
        f = m.quant_conv(h)
        side = int((f.shape[2] // P) ** 0.5)
        rows = []
        for i, fi in enumerate(f.chunk(chunks=P, dim=2)):
            fi = fi.reshape(b, -1, side, side)
            ids = m.quantizes[i].f_to_idxBl_or_fhat(fi, to_fhat=False, v_patch_nums=kw["v_patch_nums"] if multi else None)
            out[f"f{i}"] = fi.numpy()
            for si, t in enumerate(ids):
                out[f"idx{i}_{si}"] = t.reshape(b, -1).numpy()
                rows.append(t.reshape(b, -1).long() + i * V)
        rec = m.img_to_reconstructed_img(x)
    tokens = torch.cat(rows, dim=1).numpy()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), x_seed=np.int32(1234 + seed), B=np.int32(B), seed=np.int32(seed), tokens=tokens,
                        rec_sub=rec[:, :, ::4, ::4].contiguous().numpy(), n_scales=np.int32(len(kw["v_patch_nums"])),
                        meta=np.array(str(meta())), **out)
    print("wrote", name, "tokens", tokens.shape, "distinct", len(np.unique(tokens)))


TRAIN_COMMON = dict(enc_type="dinov2", dec_type="dinov2", semantic_guide="dinov2", detail_guide="none", abs_pos_embed=True,
                    encoder_model="vit_base_patch14_dinov2.lvd142m", decoder_model="vit_base_patch14_dinov2.lvd142m",
                    share_quant_resi=4, start_drop=3, sem_loss_weight=0.1, guide_type_1="class")
# the four ViT-B configs of BASELINE.json (configs[1..4]) as xqgan_train.py:285-313 builds them from the yamls
TRAIN_CASES = {
    "train_fwd_cfg2_vq8192": (dict(codebook_size=8192, codebook_embed_dim=32, v_patch_nums=[16], num_latent_tokens=256, product_quant=1,
                                   codebook_drop=0.0, half_sem=False), 4, 0.0, 0.0, 100),
    "train_fwd_cfg3_vp2_16384": (dict(codebook_size=16384, codebook_embed_dim=32, v_patch_nums=[16], num_latent_tokens=256, product_quant=2,
                                      codebook_drop=0.1, half_sem=True), 10, 0.0, 0.0, 100),
    "train_fwd_cfg4_msvr10p2_4096": (dict(codebook_size=4096, codebook_embed_dim=32, v_patch_nums=[1, 1, 2, 3, 3, 4, 5, 6, 8, 11],
                                          num_latent_tokens=121, product_quant=2, codebook_drop=0.1, half_sem=True), 10, 0.0, 0.0, 100),
    "train_fwd_cfg5_robusttok": (dict(codebook_size=4096, codebook_embed_dim=64, v_patch_nums=[16], num_latent_tokens=256, product_quant=1,
                                      codebook_drop=0.0, half_sem=False), 10, 0.5, 0.1, 100),
}


def gen_train_forward(name, kw, B, alpha, beta, delta, seed):
    """VQModel.forward in train() mode (xqgan_model.py:268-365): decoder output, codebook losses, usages, semantic loss, with
    every random draw of the pass recorded — DropPath masks (vision_transformer.py:713 rates, timm DropPath), the quantizer
    dropout depths (:274), the two draws of add_perturbation (latent_perturbation.py:21-22) — so that the mirror can replay them."""
    from oracle.det_init import det_state_dict
    from oracle import timm_shim
    import contextlib, io
    R = load_reference()
    torch.manual_seed(seed)
    m = R["VQ_models"]["VQ-16"](**dict(TRAIN_COMMON, **kw)).train()
    m.load_state_dict(det_state_dict(m.state_dict(), seed))
    x = torch.rand(B, 3, 256, 256, generator=torch.Generator().manual_seed(4321 + seed)) * 2 - 1
    draws = {"rand": [], "randint": []}
    real_rand, real_randint = torch.rand, torch.randint

    def rec_rand(*a, **k):
        t = real_rand(*a, **k)
        draws["rand"].append(t.detach().cpu().clone())
        return t

    def rec_randint(*a, **k):
        t = real_randint(*a, **k)
        draws["randint"].append((tuple(int(v) for v in a[:2] if isinstance(v, int)), t.detach().cpu().clone()))
        return t
    # the `delta` nearest codes of add_perturbation (latent_perturbation.py:20: topk(d, delta, largest=False)), values and indices, of the
    # tokens of the perturbed samples: with them a test can tell a wrong pick from a pick of the SAME distance rank on a near-tie
    topk_rec = []
    real_topk = torch.topk

    def rec_topk(*a, **k):
        out = real_topk(*a, **k)
        if k.get("largest", True) is False and len(a) >= 2 and int(a[1]) == int(delta):
            topk_rec.append((out[0].detach().cpu().clone(), out[1].detach().cpu().clone()))
        return out
    timm_shim.DropPath.RECORD = []
    torch.manual_seed(seed + 17)
    torch.rand, torch.randint, torch.topk = rec_rand, rec_randint, rec_topk
    try:
        with contextlib.redirect_stdout(io.StringIO()):          # the forward prints (alpha, beta, delta) every call (:296)
            dec, (vq, commit, ent, usages), sem, detail, dep = m(x, 0, alpha, beta, delta)
    finally:
        torch.rand, torch.randint, torch.topk = real_rand, real_randint, real_topk
        masks = timm_shim.DropPath.RECORD
        timm_shim.DropPath.RECORD = None
    P, SN = kw["product_quant"], len(kw["v_patch_nums"])
    N = B * kw["num_latent_tokens"]
    dropout_rand = [t for (_, t) in draws["randint"] if tuple(t.shape) == (B,)]
    lp_prob = [t for t in draws["rand"] if tuple(t.shape) == (N,)]
    lp_idx = [t for (_, t) in draws["randint"] if tuple(t.shape) == (N,)]
    assert len(dropout_rand) == (1 if SN > 1 else 0), (len(dropout_rand), SN)
    assert len(lp_prob) == len(lp_idx) == (1 if P == 1 else 0)
    assert len(topk_rec) == len(lp_prob)
    n_pert_tok = int(B * beta) * kw["num_latent_tokens"]
    lp_topk_val = topk_rec[0][0][:n_pert_tok].numpy().astype(np.float32) if topk_rec else np.zeros((0, 0), np.float32)
    lp_topk_idx = topk_rec[0][1][:n_pert_tok].numpy().astype(np.int32) if topk_rec else np.zeros((0, 0), np.int32)
    dec = dec.detach()
    np.savez(os.path.join(OUT, name + ".npz"), seed=np.int32(seed), B=np.int32(B), alpha=np.float32(alpha), beta=np.float32(beta),
             delta=np.int32(delta), droppath=torch.stack(masks).numpy() if masks else np.zeros((0, B), np.float32),
             dropout_rand=dropout_rand[0].numpy().astype(np.int64) if dropout_rand else np.zeros(0, np.int64),
             lp_prob=lp_prob[0].numpy() if lp_prob else np.zeros(0, np.float32),
             lp_idx=lp_idx[0].numpy().astype(np.int64) if lp_idx else np.zeros(0, np.int64),
             lp_topk_val=lp_topk_val, lp_topk_idx=lp_topk_idx,
             dec_sub=dec[:, :, ::4, ::4].contiguous().numpy(), dec_mean=np.float64(dec.double().mean()), dec_l2=np.float64(dec.double().square().mean().sqrt()),
             dec_absmax=np.float32(dec.abs().max()), vq=np.float32(float(vq)), commit=np.float32(float(commit)), entropy=np.float32(float(ent)),
             usages=np.array(usages, np.float32), sem=np.float32(float(sem)), dep=np.float32(float(dep)), meta=np.array(str(meta())))
    print("wrote", name, "masks", len(masks), "vq", float(vq), "commit", float(commit), "sem", float(sem), "dep", float(dep), "usages", [round(float(u), 2) for u in usages][:3])


# parameters whose reference gradients are recorded (sub-sampled to <= ~16k entries each): the decoder's last layer (the tensor the
# adaptive GAN weight differentiates, xqgan_train.py:452), the 1x1 convs around the quantizer, the codebook(s), first / last
# transformer block of encoder and decoder (qkv weight, fc1 bias, LayerScale), position / token tables, the Phi convs
GRAD_TAPS = ["decoder.to_pixel.model.weight", "decoder.to_pixel.model.bias", "quant_conv.weight", "quant_conv.bias", "post_quant_conv.weight",
             "quantize.embedding.weight", "quantizes.0.embedding.weight", "quantizes.1.embedding.weight",
             "quantizes.0.quant_resi.qresi_ls.0.weight", "quantizes.1.quant_resi.qresi_ls.3.weight",
             "encoder.model.blocks.0.attn.qkv.weight", "encoder.model.blocks.11.attn.qkv.weight", "encoder.model.blocks.0.mlp.fc1.bias",
             "encoder.model.blocks.11.mlp.fc2.weight", "encoder.model.blocks.5.ls1.gamma", "encoder.model.pos_embed", "encoder.latent_tokens",
             "encoder.model.patch_embed.proj.weight", "encoder.model.norm.weight",
             "decoder.model.blocks.0.attn.qkv.weight", "decoder.model.blocks.11.attn.proj.weight", "decoder.model.blocks.11.mlp.fc1.bias",
             "decoder.model.blocks.6.mlp.fc1.weight", "decoder.model.pos_embed", "decoder.mask_token", "decoder.lvl_embed.weight"]


def grad_subsample(t):
    """deterministic sub-sample of a gradient tensor: every k-th entry of the flattened tensor, k chosen for <= 16384 entries"""
    f = t.reshape(-1)
    k = max(1, (f.numel() + 16383) // 16384)
    return f[::k]


def gen_train_backward(name, fwd_name, kw, B, alpha, beta, delta, seed):
    """One reference training backward at model level (xqgan_train.py:439-462 without the GAN / LPIPS terms, which need downloaded
    checkpoints): loss = mse(recons, imgs) + vq + commit + entropy + semantic + dependency of VQModel.forward in train() mode,
    .backward() through the reference's own autograd.  Same weights, images and random draws as the forward golden `fwd_name`
    (asserted).  Recorded twice: fp32, and under torch.autocast('cpu', bfloat16) — the reference's own reduced-precision
    backward, the yardstick for the MI355X bf16 training kernels (tests/test_train_backward_parity.py)."""
    from oracle.det_init import det_state_dict
    from oracle import timm_shim
    import contextlib, io
    R = load_reference()
    fwd = np.load(os.path.join(OUT, fwd_name + ".npz"), allow_pickle=True)
    out = {}
    for tag, amp in (("f32", False), ("bf16", True)):
        torch.manual_seed(seed)
        m = R["VQ_models"]["VQ-16"](**dict(TRAIN_COMMON, **kw)).train()
        m.load_state_dict(det_state_dict(m.state_dict(), seed))
        x = torch.rand(B, 3, 256, 256, generator=torch.Generator().manual_seed(4321 + seed)) * 2 - 1
        timm_shim.DropPath.RECORD = []
        torch.manual_seed(seed + 17)
        try:
            with contextlib.redirect_stdout(io.StringIO()), torch.autocast("cpu", dtype=torch.bfloat16, enabled=amp):
                dec, (vq, commit, ent, usages), sem, detail, dep = m(x, 0, alpha, beta, delta)
                loss = torch.nn.functional.mse_loss(dec.float(), x) + vq + commit + ent + sem + dep
        finally:
            masks = timm_shim.DropPath.RECORD
            timm_shim.DropPath.RECORD = None
        if not amp:      # same pass as the forward golden: identical draws -> identical output
            assert np.array_equal(torch.stack(masks).numpy(), fwd["droppath"]), "DropPath draws differ from the forward golden"
            assert np.abs(dec.detach()[:, :, ::4, ::4].numpy() - fwd["dec_sub"]).max() <= 1e-6
        loss.backward()
        params = dict(m.named_parameters())
        out[f"loss_{tag}"] = np.float64(loss.item())
        for n in GRAD_TAPS:
            if n in params and params[n].grad is not None:
                g = params[n].grad.detach().float()
                out[f"{tag}:{n}"] = grad_subsample(g).numpy().copy()
                out[f"{tag}:{n}:l2"] = np.float64(g.double().square().sum().sqrt())
        print(name, tag, "loss", loss.item(), "taps", sum(1 for k in out if k.startswith(tag + ":") and not k.endswith(":l2")))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), fwd_name=np.array(fwd_name), taps=np.array([n for n in GRAD_TAPS if f"f32:{n}" in out]),
                        meta=np.array(str(meta())), **out)
    print("wrote", name)


# BASELINE config 1 (VQ-4096.yaml geometry on the CNN encoder / decoder — the one model the reference runs end to end here WITHOUT any
# shim: xqgan_model.py:454-704 Encoder / Decoder / ResnetBlock / AttnBlock / Up / Downsample): parameters whose reference gradients are
# recorded.  conv_in / conv_out of both halves, a GroupNorm scale and shift at three depths, the AttnBlock projections, a strided
# (Downsample) and an up-sampling conv, a nin_shortcut (1x1), the 1x1 convs around the quantizer, the codebook.
CNN_KW = dict(codebook_size=4096, codebook_embed_dim=64, v_patch_nums=[16], enc_type="cnn", dec_type="cnn", semantic_guide="none",
              detail_guide="none", num_latent_tokens=256, product_quant=1)
CNN_GRAD_TAPS = ["encoder.conv_in.weight", "encoder.conv_in.bias", "encoder.conv_blocks.0.res.0.norm1.weight", "encoder.conv_blocks.0.res.0.norm1.bias",
                 "encoder.conv_blocks.0.res.1.conv2.weight", "encoder.conv_blocks.0.downsample.conv.weight", "encoder.conv_blocks.2.res.0.nin_shortcut.weight",
                 "encoder.conv_blocks.2.res.0.conv1.weight", "encoder.conv_blocks.4.res.0.conv1.weight", "encoder.conv_blocks.4.attn.0.q.weight",
                 "encoder.conv_blocks.4.attn.1.proj_out.weight", "encoder.conv_blocks.4.attn.0.norm.weight", "encoder.mid.1.k.weight",
                 "encoder.mid.1.v.bias", "encoder.mid.2.norm2.weight", "encoder.norm_out.weight", "encoder.conv_out.weight", "encoder.conv_out.bias",
                 "quant_conv.weight", "quant_conv.bias", "quantize.embedding.weight", "post_quant_conv.weight", "post_quant_conv.bias",
                 "decoder.conv_in.weight", "decoder.mid.1.proj_out.weight", "decoder.mid.1.norm.bias", "decoder.conv_blocks.0.attn.2.proj_out.weight",
                 "decoder.conv_blocks.0.res.0.conv1.weight", "decoder.conv_blocks.0.upsample.conv.weight", "decoder.conv_blocks.1.res.0.nin_shortcut.weight",
                 "decoder.conv_blocks.2.res.1.norm2.weight", "decoder.conv_blocks.3.upsample.conv.weight", "decoder.conv_blocks.4.res.2.conv2.weight",
                 "decoder.conv_blocks.4.res.0.norm1.bias", "decoder.norm_out.weight", "decoder.norm_out.bias", "decoder.conv_out.weight", "decoder.conv_out.bias"]


def gen_train_cnn(fwd_name, bwd_name, B, alpha, beta, delta, seed):
    """Train-mode VQModel.forward (xqgan_model.py:268-365) AND one training backward (xqgan_train.py:439-462 without the GAN / LPIPS terms)
    of the reference's CNN tokenizer, BASELINE config 1 — Encoder :454-514, Decoder :518-584, ResnetBlock :587-622, AttnBlock :625-659,
    Upsample :675-686, Downsample :689-704, VectorQuantizer :745-801, add_perturbation (int(B * beta) = 0 samples at the yaml's values).
    No stochastic layer is live (dropout_p = 0, no DropPath), so there are no draws to record besides the perturbation's (replayed as
    ranks by the mirror).  Gradients twice: fp32, and under torch.autocast('cpu', bfloat16)."""
    from oracle.det_init import det_state_dict
    import contextlib, io
    R = load_reference()
    out, fwd = {}, {}
    forced = {"idx": None}
    # third leg (round 6), "bf16tf": the reference under bf16 autocast with the fp32 leg's code indices TEACHER-FORCED (torch.argmin inside
    # VectorQuantizer.forward, xqgan_model.py:766, returns the fp32 leg's pick).  At B = 4 the bf16 encoder moves a few per cent of the 1024 tokens to
    # another code, and those flips — not the arithmetic of the backward pass — are 6 - 38 % of the reference's own bf16 gradient error; with
    # the indices pinned what is left is the rounding of the forward / backward arithmetic, the yardstick the hand-written CNN kernels are held to.
    for tag, amp in (("f32", False), ("bf16", True), ("bf16tf", True)):
        torch.manual_seed(seed)
        m = R["VQ_models"]["VQ-16"](**CNN_KW).train()
        m.load_state_dict(det_state_dict(m.state_dict(), seed))
        x = torch.rand(B, 3, 256, 256, generator=torch.Generator().manual_seed(4321 + seed)) * 2 - 1
        draws = {"rand": [], "randint": []}
        real_rand, real_randint = torch.rand, torch.randint

        def rec_rand(*a, **k):
            t = real_rand(*a, **k)
            draws["rand"].append(t.detach().cpu().clone())
            return t

        def rec_randint(*a, **k):
            t = real_randint(*a, **k)
            draws["randint"].append(t.detach().cpu().clone())
            return t
        real_argmin = torch.argmin
        N_tok = B * CNN_KW["num_latent_tokens"]

        def argmin(*a, **k):
            t = real_argmin(*a, **k)
            if tuple(t.shape) == (N_tok,) and (k.get("dim", a[1] if len(a) > 1 else None) == 1):
                if tag == "f32":
                    forced["idx"] = t.detach().clone()
                elif tag == "bf16tf":
                    forced["flips"] = int((t != forced["idx"]).sum())
                    return forced["idx"].clone()
            return t
        torch.manual_seed(seed + 17)
        torch.rand, torch.randint, torch.argmin = rec_rand, rec_randint, argmin
        try:
            with contextlib.redirect_stdout(io.StringIO()), torch.autocast("cpu", dtype=torch.bfloat16, enabled=amp):
                dec, (vq, commit, ent, usages), sem, detail, dep = m(x, 0, alpha, beta, delta)
                loss = torch.nn.functional.mse_loss(dec.float(), x) + vq + commit + ent + dep
        finally:
            torch.rand, torch.randint, torch.argmin = real_rand, real_randint, real_argmin
        assert sem is None and detail is None
        N = B * CNN_KW["num_latent_tokens"]
        lp_prob = [t for t in draws["rand"] if tuple(t.shape) == (N,)]
        lp_idx = [t for t in draws["randint"] if tuple(t.shape) == (N,)]
        assert len(lp_prob) == len(lp_idx) == 1
        if not amp:
            d = dec.detach().float()
            with torch.no_grad():
                h = m.quant_conv(m.encoder(x))
                idx = m.quantize.f_to_idxBl_or_fhat(h, to_fhat=False, v_patch_nums=None)[0]
            fwd = dict(seed=np.int32(seed), B=np.int32(B), alpha=np.float32(alpha), beta=np.float32(beta), delta=np.int32(delta),
                       droppath=np.zeros((0, B), np.float32), dropout_rand=np.zeros(0, np.int64), lp_prob=lp_prob[0].numpy(),
                       lp_idx=lp_idx[0].numpy().astype(np.int64), lp_topk_val=np.zeros((0, 0), np.float32), lp_topk_idx=np.zeros((0, 0), np.int32),
                       dec_sub=d[:, :, ::4, ::4].contiguous().numpy(), dec_mean=np.float64(d.double().mean()),
                       dec_l2=np.float64(d.double().square().mean().sqrt()), dec_absmax=np.float32(d.abs().max()), vq=np.float32(float(vq)),
                       commit=np.float32(float(commit)), entropy=np.float32(float(ent)), usages=np.array(usages, np.float32),
                       sem=np.float32(0.0), dep=np.float32(float(dep)), idx=idx.numpy().astype(np.int32), f=h.numpy(), meta=np.array(str(meta())))
            assert torch.equal(idx.reshape(-1), forced["idx"].reshape(-1)), "the inference twin and the training forward picked different codes"
        loss.backward()
        params = dict(m.named_parameters())
        out[f"loss_{tag}"] = np.float64(loss.item())
        for n in CNN_GRAD_TAPS:
            g = params[n].grad.detach().float()
            out[f"{tag}:{n}"] = grad_subsample(g).numpy().copy()
            out[f"{tag}:{n}:l2"] = np.float64(g.double().square().sum().sqrt())
        # the global gradient norm as the trainer's clipping sees it (xqgan_train.py:456-458: clip_grad_norm_(vq_model.parameters(), max_grad_norm))
        out[f"gnorm_{tag}"] = np.float64(float(torch.nn.utils.clip_grad_norm_(m.parameters(), 1e30)))
        if tag == "bf16tf":
            out["bf16tf_flips_replaced"] = np.int32(forced["flips"])      # tokens whose bf16 pick differed from the forced fp32 pick
        print(bwd_name, tag, "loss", loss.item(), "vq", float(vq), "commit", float(commit), "usages", usages, "grad norm", out[f"gnorm_{tag}"],
              "" if tag != "bf16tf" else f"(teacher-forced: {forced['flips']} of {N_tok} tokens had flipped)")
    np.savez(os.path.join(OUT, fwd_name + ".npz"), **fwd)
    np.savez_compressed(os.path.join(OUT, bwd_name + ".npz"), fwd_name=np.array(fwd_name), taps=np.array(CNN_GRAD_TAPS), meta=np.array(str(meta())), **out)
    print("wrote", fwd_name, bwd_name)


def gen_model_bf16(name, kw, seed):
    """the same image through the reference under torch.autocast('cpu', bfloat16): what the reference's own reduced-precision
    path does to reconstructions and indices — the yardstick for the MI355X bf16 kernels (tests/test_model_parity.py)."""
    from oracle.det_init import det_state_dict
    R = load_reference()
    torch.manual_seed(seed)
    m = R["VQ_models"]["VQ-16"](**kw).eval()
    m.load_state_dict(det_state_dict(m.state_dict(), seed))
    x = torch.rand(1, 3, 256, 256, generator=torch.Generator().manual_seed(1234 + seed)) * 2 - 1
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
        rec = m.img_to_reconstructed_img(x).float()
        h = m.encoder(x)
        if kw["enc_type"] == "dinov2":
            b, l, c = h.shape
            h = h.view(b, int(l ** 0.5), int(l ** 0.5), c).permute(0, 3, 1, 2)
        f = m.quant_conv(h).float()
        idx = m.quantize.f_to_idxBl_or_fhat(f, to_fhat=False, v_patch_nums=None)[0]
    np.savez(os.path.join(OUT, name + "_bf16.npz"), rec_bf16=rec.numpy(), idx_bf16=idx.numpy(), f_bf16=f.numpy(), meta=np.array(str(meta())))
    print("wrote", name + "_bf16", "rec range", float(rec.min()), float(rec.max()))


def reference_vqloss(aug_prob=0.0):
    """The reference's VQLoss (vq_loss.py:79-152) constructed offline: its three network dependencies are cut, nothing else —
      * torchvision.models.vgg16 -> oracle/torchvision_shim.py (architecture restated, random init);
      * LPIPS.load_from_pretrained (lpips.py:66-69: downloads vgg.pth) -> no-op;
      * torch.hub.load_state_dict_from_url (discriminator_dino.py:178: the DINO-S checkpoint) -> empty state dict, and DinoDisc's default
        device 'cuda' -> 'cpu' (its constructor moves the frozen trunk there, :196).
    LPIPS / vgg16 wrapper / NetLinLayer / DinoDisc / DiffAug / VQLoss.forward run unmodified."""
    import importlib
    from oracle import torchvision_shim
    load_reference()
    rl = importlib.import_module("tokenizer.tokenizer_image.lpips")
    rl.models = torchvision_shim.models
    rl.LPIPS.load_from_pretrained = lambda self, name="vgg_lpips": None
    dd = importlib.import_module("tokenizer.tokenizer_image.discriminator_dino")
    d = list(dd.DinoDisc.__init__.__defaults__)
    d[1] = "cpu"
    dd.DinoDisc.__init__.__defaults__ = tuple(d)
    rv = importlib.import_module("tokenizer.tokenizer_image.vq_loss")
    saved = torch.hub.load_state_dict_from_url
    torch.hub.load_state_dict_from_url = lambda *a, **k: {}
    try:
        # xqgan_train.py:320-335 with the argparse defaults (:90 disc_weight 0.5) and the VQ-8192.yaml entries (lecam 0.001, adaptive weight)
        L = rv.VQLoss(disc_start=0, disc_weight=0.5, disc_type="dinodisc", disc_loss="hinge", gen_adv_loss="hinge", image_size=256,
                      perceptual_weight=1.0, reconstruction_weight=1.0, reconstruction_loss="l2", codebook_weight=1.0, lecam_loss_weight=0.001,
                      disc_adaptive_weight=True, norm_type="bn", aug_prob=aug_prob)
    finally:
        torch.hub.load_state_dict_from_url = saved
    return L, rv


def gen_vqloss(name, B, seed):
    """VQLoss as a whole (vq_loss.py:161-261, lpips.py:83-96,118-155, discriminator_dino.py:157-248): generator loss with the adaptive
    weight, its gradient into the reconstruction and the decoder's last layer, then the discriminator loss (hinge + LeCAM) and the gradients
    of the head parameters.  train() mode as xqgan_train.py:417, LPIPS's dropout layers in eval() (per-rank device RNG upstream), aug_prob = 0
    (DiffAug has its own same-draws test)."""
    from oracle.det_init import det_state_dict, vqloss_inputs
    L, rv = reference_vqloss(aug_prob=0.0)
    L.load_state_dict(det_state_dict(L.state_dict(), seed))
    proxy = L.discriminator.dino_proxy[0]
    proxy.load_state_dict({k[len("dino_proxy."):]: v for k, v in det_state_dict({"dino_proxy." + k: v for k, v in proxy.state_dict().items()}, seed).items()})
    L.train()
    L.perceptual_loss.eval()
    imgs, pre, last0 = vqloss_inputs(B, seed)
    pre = pre.requires_grad_(True)
    last = torch.nn.Parameter(last0.clone())
    cb = (torch.tensor(0.1), torch.tensor(0.02), torch.tensor(0.0), [1.0])
    out = {}
    rec = torch.nn.functional.conv2d(pre, last)
    # the pieces, recorded one by one (same module state: the spectral-norm power iteration advances per discriminator forward, so the
    # discriminator is evaluated exactly as often as one generator step + one discriminator step do: fake, [fake, real])
    import copy
    L2 = copy.deepcopy(L)          # for the separately recorded pieces: leaves L's spectral-norm vectors untouched
    with torch.no_grad():
        out["rec_loss"] = np.float64(L2.rec_loss(imgs, rec).item())
        out["p_loss"] = np.float64(torch.mean(L2.perceptual_loss(imgs, rec)).item())
    rec2 = torch.nn.functional.conv2d(pre, last)
    nll = L2.rec_weight * L2.rec_loss(imgs, rec2) + L2.perceptual_weight * torch.mean(L2.perceptual_loss(imgs, rec2))
    adv = L2.gen_adv_loss(L2.discriminator(L2.daug.aug(rec2, 0)))
    out["adv_loss"] = np.float64(adv.item())
    out["d_weight"] = np.float64(L2.calculate_adaptive_weight(nll, adv, last_layer=last).item())
    # the generator step as the trainer runs it (xqgan_train.py:447-456)
    loss = L(cb, None, None, 0.0, imgs, rec, optimizer_idx=0, global_step=5, last_layer=last, logger=None, log_every=1000000)
    loss.backward()
    out["gen_loss"] = np.float64(loss.item())
    out["g_pre_sub"] = grad_subsample(pre.grad).numpy().copy()
    out["g_pre_l2"] = np.float64(pre.grad.double().square().sum().sqrt())
    out["g_last"] = last.grad.numpy().copy()
    head_names = [n for n, p in L.discriminator.named_parameters() if p.requires_grad]
    # upstream's generator backward deposits head gradients too; optimizer_disc.zero_grad() discards them (xqgan_train.py:465)
    for p in L.discriminator.parameters():
        p.grad = None
    d = L(cb, None, None, 0.0, imgs, rec.detach(), optimizer_idx=1, global_step=5, logger=None, log_every=1000000)
    d.backward()
    out["disc_loss"] = np.float64(d.item())
    out["lecam_real"] = np.float64(L.lecam_ema.logits_real_ema)
    out["lecam_fake"] = np.float64(L.lecam_ema.logits_fake_ema)
    params = dict(L.discriminator.named_parameters())
    for n in head_names:
        g = params[n].grad
        out["gd:" + n] = grad_subsample(g).numpy().copy()
        out["gd:" + n + ":l2"] = np.float64(g.double().square().sum().sqrt())
    # the reference's OWN reduced-precision pass (torch.autocast('cpu', bfloat16) around both half-steps, as xqgan_train.py:447,466 wraps them
    # in torch.cuda.amp.autocast): the yardstick for the MI355X bf16 training kernels — how far bf16 moves these quantities is a property of
    # the function (a randomly initialised discriminator behind kinks), measured here instead of assumed
    Lb, _ = reference_vqloss(aug_prob=0.0)
    Lb.load_state_dict(det_state_dict(Lb.state_dict(), seed))
    pb = Lb.discriminator.dino_proxy[0]
    pb.load_state_dict({k[len("dino_proxy."):]: v for k, v in det_state_dict({"dino_proxy." + k: v for k, v in pb.state_dict().items()}, seed).items()})
    Lb.train()
    Lb.perceptual_loss.eval()
    pre_b = pre.detach().clone().requires_grad_(True)
    last_b = torch.nn.Parameter(last0.clone())
    with torch.autocast("cpu", dtype=torch.bfloat16):
        rec_b = torch.nn.functional.conv2d(pre_b, last_b)
        loss_b = Lb(cb, None, None, 0.0, imgs, rec_b, optimizer_idx=0, global_step=5, last_layer=last_b, logger=None, log_every=1000000)
    loss_b.backward()
    out["bf16:gen_loss"] = np.float64(loss_b.item())
    out["bf16:g_pre_sub"] = grad_subsample(pre_b.grad.float()).numpy().copy()
    out["bf16:g_last"] = last_b.grad.float().numpy().copy()
    for p in Lb.discriminator.parameters():
        p.grad = None
    with torch.autocast("cpu", dtype=torch.bfloat16):
        d_b = Lb(cb, None, None, 0.0, imgs, rec_b.detach(), optimizer_idx=1, global_step=5, logger=None, log_every=1000000)
    d_b.backward()
    out["bf16:disc_loss"] = np.float64(d_b.item())
    pbn = dict(Lb.discriminator.named_parameters())
    for n in head_names:
        out["bf16:gd:" + n] = grad_subsample(pbn[n].grad.float()).numpy().copy()
    print("reference bf16 autocast: gen_loss", loss_b.item(), "disc_loss", d_b.item(), "| d loss / d pre rel. distance to fp32:",
          float(np.linalg.norm(out["bf16:g_pre_sub"] - out["g_pre_sub"]) / np.linalg.norm(out["g_pre_sub"])))
    # conditioning of the recorded gradient: the discriminator path has kinks (LeakyReLU heads behind batch statistics) — relative input
    # noise at the fp32 rounding level moves the REFERENCE's own d adv / d recons in jumps of ~0.15 %.  Recorded so that the tests' bounds
    # on that gradient (and on the adaptive weight that is a ratio of its norms) are derived from a measurement, not chosen.
    def adv_grad(noise, s):
        Lc, _ = reference_vqloss(aug_prob=0.0)
        Lc.load_state_dict(det_state_dict(Lc.state_dict(), seed))
        pc = Lc.discriminator.dino_proxy[0]
        pc.load_state_dict({k[len("dino_proxy."):]: v for k, v in det_state_dict({"dino_proxy." + k: v for k, v in pc.state_dict().items()}, seed).items()})
        Lc.train()
        r = rec.detach() * (1 + noise * torch.randn(rec.shape, generator=torch.Generator().manual_seed(s)))
        r.requires_grad_(True)
        return torch.autograd.grad(Lc.gen_adv_loss(Lc.discriminator(Lc.daug.aug(r, 0))), r)[0]
    g0 = adv_grad(0.0, 0)
    out["adv_grad_rel_change_under_noise"] = np.array([[n, float(((adv_grad(n, s) - g0).norm() / g0.norm()).item())]
                                                       for n, s in ((1e-7, 1), (1e-7, 2), (1e-6, 3), (1e-6, 4), (1e-5, 5), (1e-5, 6))])
    print("conditioning (input noise, relative change of d adv / d recons):", out["adv_grad_rel_change_under_noise"].tolist())
    np.savez_compressed(os.path.join(OUT, name + ".npz"), B=np.int64(B), seed=np.int64(seed), head_names=np.array(head_names),
                        meta=np.array(str(meta())), **out)
    print("wrote", name, {k: float(v) for k, v in out.items() if np.ndim(v) == 0 and not k.startswith('gd:')})


def main():
    os.makedirs(OUT, exist_ok=True)
    only = sys.argv[1] if len(sys.argv) > 1 else ""
    if only in ("model", ""):
        # BASELINE config 1 (the reference's own CPU-runnable case): VQ-4096, CNN encoder/decoder, 72 M parameters
        gen_model("model_cfg1_cnn_vq4096", dict(codebook_size=4096, codebook_embed_dim=64, v_patch_nums=[16], enc_type="cnn",
                                                dec_type="cnn", semantic_guide="none", detail_guide="none",
                                                num_latent_tokens=256, product_quant=1), seed=31)
        # BASELINE config 2 geometry: VQ-8192, DINOv2 ViT-B encoder/decoder (vendored reference ViT over the timm shim)
        gen_model("model_cfg2_vitb_vq8192", dict(codebook_size=8192, codebook_embed_dim=32, v_patch_nums=[16], enc_type="dinov2",
                                                 dec_type="dinov2", semantic_guide="none", detail_guide="none",
                                                 num_latent_tokens=256, product_quant=1, abs_pos_embed=True,
                                                 encoder_model="vit_base_patch14_dinov2.lvd142m",
                                                 decoder_model="vit_base_patch14_dinov2.lvd142m"), seed=32)
        if only:
            return
    if only in ("vqloss", ""):
        gen_vqloss("vqloss_dinodisc_b4", 4, seed=61)
        if only:
            return
    if only in ("lfq", ""):
        # MSBR10P2-4096 geometry (12 bit channels, 1x1 -> 11x11 ladder, codebook_l2_norm, codebook_drop 0.1, start_drop 3)
        gen_lfq("lfq_msbr_c12_11grid_b6", 12, 6, [1, 1, 2, 3, 3, 4, 5, 6, 8, 11], seed=40)
        # no z-norm, 14 bits (MSBR10P2-16384 width), heavier dropout, 16x16 grid (skips the last area-pool)
        gen_lfq("lfq_raw_c14_16grid_b4", 14, 4, [1, 2, 3, 4, 6, 8, 11, 16], seed=41, using_znorm=False, codebook_drop=0.5, start_drop=2)
        if only:
            return
    if only in ("msvq", ""):
        # BASELINE config 4 ladder (MSVR10P2: 1x1 -> 11x11, C=32, codebook_drop 0.1, start_drop 3); V reduced to keep the
        # fixture small (V=4096 full-size runs are oracle-vs-HIP tests)
        gen_msvq("msvq_cfg4_ladder_v1024_c32_b12", 1024, 32, 12, [1, 1, 2, 3, 3, 4, 5, 6, 8, 11], seed=20)
        gen_msvq("msvq_16grid_v512_c16_b4", 512, 16, 4, [1, 2, 3, 4, 5, 6, 8, 10, 13, 16], seed=21, codebook_drop=0.5, start_drop=1)
        gen_msvq("msvq_rawl2_v256_c8_b3", 256, 8, 3, [1, 2, 4, 7], seed=22, using_znorm=False, codebook_drop=0.4, start_drop=2)
        gen_msvq("msvq_var_models_quant_v512_c32_b4", 512, 32, 4, [1, 2, 3, 4, 5, 6, 8, 10], seed=23, var_variant=True)
        if only:
            return
    if only in ("train", "train5"):
        for i, (nm, (kw, B, al, be, de)) in enumerate(TRAIN_CASES.items()):
            if only == "train5" and "cfg5" not in nm:
                continue
            gen_train_forward(nm, kw, B, al, be, de, seed=60 + i)
        return
    if only == "tokens":
        base = dict(enc_type="dinov2", dec_type="dinov2", semantic_guide="none", detail_guide="none", abs_pos_embed=True,
                    encoder_model="vit_base_patch14_dinov2.lvd142m", decoder_model="vit_base_patch14_dinov2.lvd142m", share_quant_resi=4)
        # BASELINE config 3 (VP2-16384: two single-scale quantizers) and config 4 (MSVR10P2-4096: two 10-scale ladders)
        gen_tokens("tokens_cfg3_vp2_16384", dict(base, codebook_size=16384, codebook_embed_dim=32, v_patch_nums=[16], num_latent_tokens=256,
                                                 product_quant=2, half_sem=True), 2, seed=70)
        gen_tokens("tokens_cfg4_msvr10p2_4096", dict(base, codebook_size=4096, codebook_embed_dim=32, v_patch_nums=[1, 1, 2, 3, 3, 4, 5, 6, 8, 11],
                                                     num_latent_tokens=121, product_quant=2, half_sem=True), 2, seed=71)
        return
    if only == "traincnn":
        # BASELINE config 1: train-mode forward + backward of the reference's CNN tokenizer, unshimmed (yaml / CLI defaults alpha = beta = 0, delta = 100)
        gen_train_cnn("train_fwd_cfg1_cnn_vq4096", "train_bwd_cfg1_cnn_vq4096", 4, 0.0, 0.0, 100, seed=59)
        return
    if only == "trainbwd":
        for i, (nm, (kw, B, al, be, de)) in enumerate(TRAIN_CASES.items()):
            gen_train_backward(nm.replace("train_fwd_", "train_bwd_"), nm, kw, B, al, be, de, seed=60 + i)
        return
    if only == "bf16":
        gen_model_bf16("model_cfg1_cnn_vq4096", dict(codebook_size=4096, codebook_embed_dim=64, v_patch_nums=[16], enc_type="cnn",
                                                     dec_type="cnn", semantic_guide="none", detail_guide="none",
                                                     num_latent_tokens=256, product_quant=1), seed=31)
        gen_model_bf16("model_cfg2_vitb_vq8192", dict(codebook_size=8192, codebook_embed_dim=32, v_patch_nums=[16], enc_type="dinov2",
                                                      dec_type="dinov2", semantic_guide="none", detail_guide="none",
                                                      num_latent_tokens=256, product_quant=1, abs_pos_embed=True,
                                                      encoder_model="vit_base_patch14_dinov2.lvd142m",
                                                      decoder_model="vit_base_patch14_dinov2.lvd142m"), seed=32)
        return
    if only in ("lfqvar", ""):
        gen_lfq_var_helpers("vh_lfq_c12_11grid_b3", 12, 3, [1, 1, 2, 3, 3, 4, 5, 6, 8, 11], seed=53)
        gen_lfq_var_helpers("vh_lfq_raw_c14_16grid_b2", 14, 2, [1, 2, 3, 4, 6, 8, 11, 16], seed=54, using_znorm=False)
        if only:
            return
    if only in ("var", ""):
        # VAR-d16 geometry (1x1 -> 16x16, 10 scales, 4 partially shared Phi) and the MSVR10P2 ladder; models/quant.py twin
        gen_var_helpers("var_helpers_16grid_v512_c16_b3", 512, 16, 3, [1, 2, 3, 4, 5, 6, 8, 10, 13, 16], seed=50)
        gen_var_helpers("var_helpers_cfg4_ladder_v1024_c32_b4", 1024, 32, 4, [1, 1, 2, 3, 3, 4, 5, 6, 8, 11], seed=51)
        gen_var_helpers("var_helpers_models_quant_v256_c8_b2", 256, 8, 2, [1, 2, 3, 4, 6, 8], seed=52, var_variant=True, share=1)
        if only:
            return
    if only == "perturb":
        gen_perturb("perturb_v1024_c64_b8", 1024, 64, 8, 8, 8, seed=5, alpha=0.5, beta=0.25, delta=100)
        gen_perturb("perturb_alpha1_v512_c32_b4", 512, 32, 4, 4, 4, seed=6, alpha=1.0, beta=0.5, delta=50)
        gen_perturb("perturb_identity_v256_c16_b4", 256, 16, 4, 4, 4, seed=7, alpha=0.5, beta=0.1, delta=100)
        gen_perturb("perturb_raw_v300_c8_b3", 300, 8, 3, 5, 5, seed=8, alpha=0.7, beta=0.7, delta=20, codebook_norm=False)
        return
    # BASELINE config 1 shape (VQ-4096, C=64, B=4, 16x16) — the reference's own CPU-runnable case
    gen_vq("vq_cfg1_v4096_c64_b4", 4096, 64, 4, 16, 16, seed=0)
    # config 2 codebook geometry at a fixture-sized batch (VQ-8192, C=32)
    gen_vq("vq_cfg2_v8192_c32_b2", 8192, 32, 2, 16, 16, seed=1)
    # ragged: N not a multiple of the 32-token tile, V not a multiple of the 128-code stage, tiny C
    gen_vq("vq_ragged_v1000_c8_b3_5x7", 1000, 8, 3, 5, 7, seed=2)
    # no codebook norm (raw L2 path), C=16
    gen_vq("vq_raw_v512_c16_b2", 512, 16, 2, 8, 8, seed=3, codebook_norm=False, z_scale=0.02)
    # clustered latents: many near-duplicates -> histogram contention + near ties
    gen_vq("vq_clustered_v2048_c32_b2", 2048, 32, 2, 16, 16, seed=4, clustered=True)
    # RobustTok (config 5 geometry V=4096,C=64 is 1 MB of codebook; use V=1024 here, full size is covered
    # by oracle-vs-HIP tests): beta=0.25 of B=8 -> 2 perturbed samples, alpha=0.5, delta=100
    gen_perturb("perturb_v1024_c64_b8", 1024, 64, 8, 8, 8, seed=5, alpha=0.5, beta=0.25, delta=100)
    gen_perturb("perturb_alpha1_v512_c32_b4", 512, 32, 4, 4, 4, seed=6, alpha=1.0, beta=0.5, delta=50)
    gen_perturb("perturb_identity_v256_c16_b4", 256, 16, 4, 4, 4, seed=7, alpha=0.5, beta=0.1, delta=100)  # int(4*0.1)=0
    gen_perturb("perturb_raw_v300_c8_b3", 300, 8, 3, 5, 5, seed=8, alpha=0.7, beta=0.7, delta=20, codebook_norm=False)


if __name__ == "__main__":
    main()
