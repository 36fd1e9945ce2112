"""In-loop reconstruction FID (reference xqgan_train.py:516-567 + evaluator.py:72-115,151-191).

The reference gathers every reconstruction and every input image of the validation set to rank 0 as uint8 arrays
(dist.nn.all_gather per batch, xqgan_train.py:530-533), runs the TF-Inception graph there while the other ranks wait at a
barrier, and forms mean / covariance with numpy.  Here the statistics are built where the images are:

  * FeatureStats — streaming n, sum x, sum x x^T in fp64 on the GPU (one [D x B] x [B x D] product per batch), reduced
    across ranks with ONE all-reduce of (1 + D + D^2) doubles at the end; finalize() gives numpy.mean / numpy.cov
    (rowvar=False, N-1 normalisation: evaluator.compute_statistics :186-189) of the concatenated activations;
  * frechet_distance — evaluator.FIDStatistics.frechet_distance (:72-115) restated: |mu1-mu2|^2 + tr S1 + tr S2 - 2 tr sqrtm(S1 S2)
    with scipy's sqrtm on the host (the D x D problem is the same size for any number of images), eps fallback and
    imaginary-part check as upstream; frechet_distance_device is the same quantity from the symmetric eigenproblem
    of S1^1/2 S2 S1^1/2 on the GPU (no host round trip inside the train loop);
  * ReconstructionFID — the loop body of xqgan_train.py:521-533: reconstruct, quantise both images to uint8 exactly as :526-527
    (clamp(127.5 x + 128, 0, 255) -> uint8), push both through a frozen feature network, update two FeatureStats.
The Inception graph itself (a TF .pb download, evaluator.py:20) is not reproducible offline: `feature_fn` is any frozen
module mapping (B, 3, H, W) float images in [0, 255] to (B, D) features — the parity claim tested here is about the
reconstructions (reference CPU path vs MI355X), for which any fixed feature map serves.
"""
import warnings
from typing import Callable, Optional

import numpy as np
import torch
import torch.distributed as dist


class FeatureStats:
    """Streaming first and second moments of (N, D) activations; fp64 on the activations' device."""

    def __init__(self, dim: int, device=None):
        self.dim = dim
        self.n = torch.zeros((), dtype=torch.float64, device=device)
        self.sum = torch.zeros(dim, dtype=torch.float64, device=device)
        self.outer = torch.zeros(dim, dim, dtype=torch.float64, device=device)

    @torch.no_grad()
    def update(self, feats: torch.Tensor):
        f = feats.detach().reshape(feats.shape[0], -1).to(torch.float64)
        assert f.shape[1] == self.dim
        self.n += f.shape[0]
        self.sum += f.sum(0)
        self.outer.addmm_(f.t(), f)
        return self

    def all_reduce(self, group=None):
        """one SUM all-reduce of the packed (n, sum, outer) — replaces the per-batch image all_gather of the reference"""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            packed = torch.cat([self.n.reshape(1), self.sum, self.outer.reshape(-1)])
            dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
            self.n, self.sum, self.outer = packed[0], packed[1:1 + self.dim].clone(), packed[1 + self.dim:].reshape(self.dim, self.dim).clone()
        return self

    def finalize(self):
        """(mu, sigma) as evaluator.compute_statistics: np.mean(axis=0), np.cov(rowvar=False)"""
        n = float(self.n.item())
        mu = self.sum / n
        sigma = (self.outer - n * torch.outer(mu, mu)) / (n - 1.0)
        return mu, sigma


def frechet_distance(mu1, sigma1, mu2, sigma2, eps: float = 1e-6) -> float:
    """evaluator.py:72-115 (itself the TTUR implementation), on the host in fp64."""
    from scipy import linalg
    mu1, mu2 = np.atleast_1d(np.asarray(mu1, np.float64)), np.atleast_1d(np.asarray(mu2, np.float64))
    sigma1, sigma2 = np.atleast_2d(np.asarray(sigma1, np.float64)), np.atleast_2d(np.asarray(sigma2, np.float64))
    assert mu1.shape == mu2.shape, f"Training and test mean vectors have different lengths: {mu1.shape}, {mu2.shape}"
    assert sigma1.shape == sigma2.shape, f"Training and test covariances have different dimensions: {sigma1.shape}, {sigma2.shape}"
    diff = mu1 - mu2
    covmean, _ = linalg.sqrtm(sigma1.dot(sigma2), disp=False)
    if not np.isfinite(covmean).all():
        warnings.warn("fid calculation produces singular product; adding %s to diagonal of cov estimates" % eps)
        offset = np.eye(sigma1.shape[0]) * eps
        covmean = linalg.sqrtm((sigma1 + offset).dot(sigma2 + offset))
    if np.iscomplexobj(covmean):
        if not np.allclose(np.diagonal(covmean).imag, 0, atol=1e-3):
            raise ValueError("Imaginary component {}".format(np.max(np.abs(covmean.imag))))
        covmean = covmean.real
    return float(diff.dot(diff) + np.trace(sigma1) + np.trace(sigma2) - 2 * np.trace(covmean))


@torch.no_grad()
def frechet_distance_device(mu1: torch.Tensor, sigma1: torch.Tensor, mu2: torch.Tensor, sigma2: torch.Tensor) -> torch.Tensor:
    """The same distance without leaving the device: tr sqrtm(S1 S2) = sum_i sqrt(lambda_i(S1^1/2 S2 S1^1/2)) (the two
    products are similar matrices for positive semi-definite S1), two symmetric eigen-decompositions in fp64."""
    mu1, mu2, s1, s2 = (t.to(torch.float64) for t in (mu1, mu2, sigma1, sigma2))
    w, v = torch.linalg.eigh((s1 + s1.t()) * 0.5)
    r = (v * w.clamp_min(0).sqrt()) @ v.t()                     # S1^1/2
    m = r @ s2 @ r
    lam = torch.linalg.eigvalsh((m + m.t()) * 0.5).clamp_min(0)
    d = mu1 - mu2
    return d.dot(d) + torch.trace(s1) + torch.trace(s2) - 2.0 * lam.sqrt().sum()


def to_uint8_like_reference(x: torch.Tensor) -> torch.Tensor:
    """xqgan_train.py:526-527: clamp(127.5 x + 128, 0, 255) -> uint8 (truncation), NCHW kept"""
    return torch.clamp(127.5 * x + 128.0, 0, 255).to(torch.uint8)


@torch.no_grad()
def reconstruct_for_fid(model, x: torch.Tensor, perturb=None) -> torch.Tensor:
    """The reconstruction the FID is taken of.
      * rFID (perturb = None): `vq_model.img_to_reconstructed_img(x)`, the in-loop evaluation of xqgan_train.py:523-525;
      * pFID (RobustTok, BASELINE config 5 "pFID eval on"; README.md:57-59): perturb = (alpha, beta, delta) — the image decoded from
        PERTURBED latents.  The reference has no evaluator for it: the perturbation lives in the P = 1 branch of VQModel.forward
        (xqgan_model.py:292-297) and that forward cannot run in eval() mode upstream (VectorQuantizer.forward leaves `codebook_usage`
        unbound outside train(), :773-801).  This is that branch composed from the reference's INFERENCE entry points, in eval mode (no
        DropPath, no usage statistics):  h = encode(x);  z_q = quantize.f_to_idxBl_or_fhat(h, to_fhat=True)[0];
        z_p = add_perturbation(h, z_q, z_channels, codebook_norm, embedding, alpha, beta, delta)  (the first int(B * beta) samples of the
        batch are perturbed, latent_perturbation.py:32: beta = 1 perturbs every sample);  decode(z_p).clamp(-1, 1) as :399."""
    if perturb is None:
        return model.img_to_reconstructed_img(x)
    from .latent_perturbation import add_perturbation
    alpha, beta, delta = perturb
    if getattr(model, "product_quant", 1) != 1 or len(getattr(model, "v_patch_nums", [16])) != 1:
        raise ValueError("pFID: VQModel.forward perturbs the latents of single-quantizer, single-scale models only (xqgan_model.py:276-297)")
    was_training = model.training
    model.eval()
    try:
        q = model.quantize
        h = model.encode(x)
        zq = q.f_to_idxBl_or_fhat(h, to_fhat=True, v_patch_nums=None)[0]
        zp = add_perturbation(h.float(), zq.float(), q.z_channels, q.codebook_norm, q.embedding, alpha, beta, delta)
        dec = model.decode(zp)
    finally:
        model.train(was_training)
    return dec.float().clamp_(-1, 1)


class ReconstructionFID:
    """rFID between the inputs and their reconstructions under a frozen feature network.

        ev = ReconstructionFID(feature_fn, dim)
        for x, _ in val_loader: ev.update(x, vq_model.img_to_reconstructed_img(x))      # every rank, its own shard
        fid = ev.compute()                                                              # one all-reduce, same value on all ranks
    """

    def __init__(self, feature_fn: Callable[[torch.Tensor], torch.Tensor], dim: int, device=None, group=None):
        self.feature_fn = feature_fn
        self.ref = FeatureStats(dim, device)
        self.smp = FeatureStats(dim, device)
        self.group = group

    @torch.no_grad()
    def update(self, x: torch.Tensor, recon: torch.Tensor):
        gt = to_uint8_like_reference(x).float()
        sample = to_uint8_like_reference(recon).float()
        self.ref.update(self.feature_fn(gt))
        self.smp.update(self.feature_fn(sample))
        return self

    @torch.no_grad()
    def update_from_model(self, model, x: torch.Tensor, perturb=None):
        """rFID (perturb = None) or pFID (perturb = (alpha, beta, delta)) of `model` on the batch x: see reconstruct_for_fid"""
        return self.update(x, reconstruct_for_fid(model, x, perturb))

    def compute(self, on_device: bool = False) -> float:
        self.ref.all_reduce(self.group)
        self.smp.all_reduce(self.group)
        m1, s1 = self.smp.finalize()
        m2, s2 = self.ref.finalize()
        if on_device:
            return float(frechet_distance_device(m1, s1, m2, s2).item())
        return frechet_distance(m1.cpu().numpy(), s1.cpu().numpy(), m2.cpu().numpy(), s2.cpu().numpy())
