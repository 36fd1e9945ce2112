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


# ---- op-level stand-ins: the product MODULES (imagefolder_amd.xqgan_model.VectorQuantizer, quant.VectorQuantizer2, VQModel, their
#      histogram all-reduce, usage EMA, loss assembly) run unchanged on the host; only the three autograd ops that exist as HIP
#      kernels alone are swapped for the ATen restatements.  Used by the multi-process (gloo) CPU tests of the train step. ---------
class _Op:
    def __init__(self, fn):
        self.apply = fn


def _vq_apply(z, weight, beta, codebook_norm):
    zq, idx, vq, commit, hist = tr.vq_forward(z.float(), weight, beta, codebook_norm)
    return zq, vq, commit, idx, hist


def _msvq_apply(f, weight, phi_w, phi_b, n_quant, cfg):
    return tr.msvq_ladder(f.float(), weight, phi_w, phi_b, n_quant, cfg["patch_nums"], cfg["phi_sel"], cfg["phi_ratio"],
                          cfg["using_znorm"], cfg["skip_last_pool"])


def _perturb_apply(z, z_q, weight, codebook_norm, n_pert, rank):
    if n_pert == 0:
        return z_q
    B = z.shape[0]
    # the ranks are already drawn (latent_perturbation.draw_ranks): random_prob = 0 > alpha = 1 never holds, so tr.perturb picks
    # random_idx = rank for every token; top-(max rank + 1) holds every rank that is asked for
    out, _ = tr.perturb(z.float(), z_q, weight, codebook_norm, 1.0, (n_pert + 0.5) / B, int(rank.max().item()) + 1,
                        torch.zeros(rank.shape), rank)
    return out


class install_ops:
    """with install_ops(): imagefolder_amd.ops.{VQStraightThrough, MSVQLadder, PerturbStraightThrough} -> host restatements"""

    def __enter__(self):
        from imagefolder_amd import ops
        self.ops = ops
        self.saved = (ops.VQStraightThrough, ops.MSVQLadder, ops.PerturbStraightThrough)
        ops.VQStraightThrough, ops.MSVQLadder, ops.PerturbStraightThrough = _Op(_vq_apply), _Op(_msvq_apply), _Op(_perturb_apply)
        return self

    def __exit__(self, *a):
        self.ops.VQStraightThrough, self.ops.MSVQLadder, self.ops.PerturbStraightThrough = self.saved
