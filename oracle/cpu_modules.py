"""TEST / BASELINE INFRASTRUCTURE — host (ATen, fp32) stand-ins for the two GPU-only quantizer ops, so that bench.py's
`cpu_baseline` leg can time the COMPLETE train step on the host cores with the reference's own expressions
(oracle/torch_restatement.py restates xqgan_model.py:745-801 and latent_perturbation.py:4-35).  The product path never
imports this module: imagefolder_amd's quantizer ops raise on CPU tensors by design."""
import torch
from torch import nn

from . import torch_restatement as tr


class CpuVectorQuantizer(nn.Module):
    """VectorQuantizer.forward (xqgan_model.py:745-801) on ATen CPU ops; same return tuple as the mirror."""

    def __init__(self, src):
        super().__init__()
        self.embedding = src.embedding
        self.vocab_size, self.z_channels, self.beta, self.codebook_norm = src.vocab_size, src.z_channels, src.beta, src.codebook_norm
        self.register_buffer("ema_vocab_hit_SV", src.ema_vocab_hit_SV.clone())
        self.record_hit = 0

    def forward(self, z, ret_usages=True, dropout=None):
        zq, idx, vq, commit, hist = tr.vq_forward(z, self.embedding.weight, self.beta, self.codebook_norm)
        if self.record_hit == 0:
            self.ema_vocab_hit_SV.copy_(hist)
        else:
            self.ema_vocab_hit_SV.mul_(0.9).add_(hist.mul(0.1))
        self.record_hit += 1
        margin = (z.numel() / self.z_channels) / self.vocab_size * 0.08
        usage = (self.ema_vocab_hit_SV >= margin).float().mean().item() * 100
        return zq, [usage], vq, commit, 0.0


def cpu_add_perturbation(z, z_q, z_channels, codebook_norm, codebook, alpha, beta, delta):
    """add_perturbation (latent_perturbation.py:4-35) with its two RNG draws made on the host"""
    N = z.shape[0] * z.shape[2] * z.shape[3]
    out, _ = tr.perturb(z, z_q, codebook.weight, codebook_norm, alpha, beta, delta, torch.rand(N), torch.randint(0, delta, (N,)))
    return out


def install(model):
    """swap the GPU-only quantizer pieces of an imagefolder_amd VQModel (built on CPU) for the host stand-ins"""
    from imagefolder_amd import xqgan_model
    if not isinstance(model.quantize, xqgan_model.VectorQuantizer):
        raise NotImplementedError("host stand-in exists for the single-scale VectorQuantizer only")
    model.quantize = CpuVectorQuantizer(model.quantize)
    orig_cls = xqgan_model.VectorQuantizer

    class _Patch:
        def __enter__(self):
            self.saved = (xqgan_model.add_perturbation, xqgan_model.VectorQuantizer)
            xqgan_model.add_perturbation = cpu_add_perturbation
            xqgan_model.VectorQuantizer = CpuVectorQuantizer      # the isinstance test in VQModel.forward
            return self

        def __exit__(self, *a):
            xqgan_model.add_perturbation, xqgan_model.VectorQuantizer = self.saved
    del orig_cls
    return _Patch()
