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
