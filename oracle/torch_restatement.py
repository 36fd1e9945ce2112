"""TEST INFRASTRUCTURE — the reference's quantizer expressions restated with the same ATen CPU ops
(fp32), for (i) timing the "reference CPU path" on the GPU box's host cores where /root/reference
does not exist (bench.py cpu_baseline, kind="port") and (ii) cross-checking the C oracle.

Each function follows the reference line by line in *meaning* (not text):
  vq_forward      xqgan_model.py:745-801   (VectorQuantizer.forward, training branch)
  perturb         latent_perturbation.py:4-35 (explicit RNG draws instead of device RNG)
"""
import torch
import torch.nn.functional as F


def _flatten_tokens(z):  # (B,C,H,W) -> (B,H,W,C), (N,C)      xqgan_model.py:750-751
    zt = z.permute(0, 2, 3, 1).contiguous()
    return zt, zt.view(-1, z.shape[1])


def distances(z_flat, emb):  # xqgan_model.py:761-763
    return z_flat.pow(2).sum(1, keepdim=True) + emb.pow(2).sum(1) - 2 * (z_flat @ emb.t())


def vq_forward(z, weight, beta=0.25, codebook_norm=True):
    """returns (z_q NCHW straight-through, idx, vq_loss, commit_loss, hist)"""
    zt, zf = _flatten_tokens(z)
    if codebook_norm:  # :753-756
        zt = F.normalize(zt, p=2, dim=-1)
        zf = F.normalize(zf, p=2, dim=-1)
        emb = F.normalize(weight, p=2, dim=-1)
    else:
        emb = weight
    idx = torch.argmin(distances(zf, emb), dim=1)  # :766
    zq = weight[idx].view(zt.shape)  # :769
    if codebook_norm:
        zq = F.normalize(zq, p=2, dim=-1)  # :771
    hist = torch.bincount(idx, minlength=weight.shape[0]).float()  # :774
    commit = beta * torch.mean((zq.detach() - zt) ** 2)  # :792
    vq = torch.mean((zq - zt.detach()) ** 2)  # :793
    out = zt + (zq - zt).detach()  # :796
    return out.permute(0, 3, 1, 2), idx, vq, commit, hist


def perturb(z, z_q, weight, codebook_norm, alpha, beta, delta, random_prob, random_idx):
    """add_perturbation with the RNG draws passed in (latent_perturbation.py:20-23 draws them on device)."""
    zt, zf = _flatten_tokens(z)
    if codebook_norm:
        zt = F.normalize(zt, p=2, dim=-1)
        zf = F.normalize(zf, p=2, dim=-1)
        emb = F.normalize(weight, p=2, dim=-1)
    else:
        emb = weight
    d = distances(zf, emb)
    _, cand = torch.topk(d, delta, dim=1, largest=False)  # :20
    ridx = torch.where(random_prob > alpha, torch.zeros_like(random_idx), random_idx)  # :23
    pick = cand[torch.arange(cand.size(0)), ridx]  # :24
    pz = weight[pick].view(zt.shape)
    if codebook_norm:
        pz = F.normalize(pz, p=2, dim=-1)
    pz = (zt + (pz - zt).detach()).permute(0, 3, 1, 2)  # :29-30
    mask = (torch.arange(z.shape[0]) < int(z.shape[0] * beta))[:, None, None, None]  # :32-33
    return torch.where(mask, pz, z_q), pick


def msvq_ladder(f, weight, phi_w, phi_b, n_quant, patch_nums, phi_sel, phi_ratio, using_znorm, skip_last_pool):
    """VectorQuantizer2.forward's ladder (tokenizer_image/quant.py:64-135) on ATen CPU ops, returned in the shape of
    imagefolder_amd.ops.MSVQLadder: (f_hat straight-through, sq_vq (SN,), sq_commit (SN,), idx_all, hist (SN,V)).
    sq_vq[s] = sum(mask_s (f_hat_s - sg f)^2) carries the gradient into the codebook / Phi convs (:131), sq_commit[s] =
    sum(mask_s (sg f_hat_s - f)^2) the one into f (:132); the caller divides by ratio_s, numel and SN as the reference does.
    phi_w (K,C,C,3,3) / phi_b (K,C): the stacked Phi convs, phi_sel[s] the one scale s uses (quant.py:110-113,280-288)."""
    B, C, H, W = f.shape
    SN = len(patch_nums)
    V = weight.shape[0]
    f_no_grad = f.detach()
    f_rest = f_no_grad.clone()                                           # :68-70
    f_hat = torch.zeros_like(f_rest)
    nq = torch.full((B,), float(SN + 1)) if n_quant is None else n_quant.float()
    sq_vq, sq_commit, idxs, hists = [], [], [], []
    E = weight.detach()
    for si, pn in enumerate(patch_nums):
        pooled = not (si == SN - 1 and skip_last_pool)                   # :91-92
        rest = (F.interpolate(f_rest, size=(pn, pn), mode='area') if pooled else f_rest).permute(0, 2, 3, 1).reshape(-1, C)
        if using_znorm:                                                  # :93-94
            idx = torch.argmax(F.normalize(rest, dim=-1) @ F.normalize(E.T, dim=0), dim=1)
        else:                                                            # :96-101
            d = rest.square().sum(1, keepdim=True) + E.square().sum(1)
            d = d.addmm(rest, E.T, alpha=-2, beta=1)
            idx = torch.argmin(d, dim=1)
        hists.append(idx.bincount(minlength=V).float())                  # :102
        idxs.append(idx)
        h = weight[idx.view(B, pn, pn)].permute(0, 3, 1, 2)              # :106-109 (gradient reaches the codebook rows)
        if si != SN - 1:
            h = F.interpolate(h, size=(H, W), mode='bicubic')
        h = h.contiguous()
        if phi_w is not None:                                            # :110-113, Phi.forward :266-268
            k = phi_sel[si]
            h = h * (1 - phi_ratio) + F.conv2d(h, phi_w[k], phi_b[k], padding=1) * phi_ratio
        mask = (torch.full((B,), float(si)) < nq)[:, None, None, None].to(h.dtype)     # :115
        f_hat = f_hat + h * mask                                         # :116
        f_rest = f_rest - h.detach()                                     # :118
        sq_vq.append(((f_hat - f_no_grad).square() * mask).sum())        # :131 (numerator)
        sq_commit.append(((f_hat.detach() - f).square() * mask).sum())   # :132 (numerator)
    f_hat_ste = (f_hat.detach() - f_no_grad) + f                         # :135
    return f_hat_ste, torch.stack(sq_vq), torch.stack(sq_commit), torch.cat(idxs), torch.stack(hists)
