"""Host-side mirror of the reference's tokenizer model API over the MI355X kernels.

Mirrors (same names, ctor signatures, parameter/buffer names, return tuples):
    VectorQuantizer   reference tokenizer/tokenizer_image/xqgan_model.py:722-833
The arithmetic runs in libxq_ops.so (include/xq_ops.h); this file keeps only the module state the
reference keeps in Python (embedding parameter, EMA hit buffer, record_hit counter).
"""
from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.distributed as tdist

from . import ops


def _dist_ready() -> bool:
    return tdist.is_available() and tdist.is_initialized()


class VectorQuantizer(nn.Module):
    """Drop-in for reference `VectorQuantizer` (xqgan_model.py:722-833).

    Differences that are deliberate and documented in DESIGN.md:
      * the N x V distance matrix is never materialised (fused fp32-MFMA kernel);
      * the usage-histogram all-reduce is skipped when no process group exists (the reference
        requires one); with a group it is the same SUM all-reduce (xqgan_model.py:775-776).
    """

    def __init__(self, vocab_size=8192, z_channels=32, beta=0.25, codebook_norm=True):
        super().__init__()
        self.vocab_size = vocab_size
        self.z_channels = z_channels
        self.beta = beta
        self.codebook_norm = codebook_norm

        # same init as the reference (:734-737): U(-1/V, 1/V) then row-l2-normalise
        self.embedding = nn.Embedding(self.vocab_size, self.z_channels)
        self.embedding.weight.data.uniform_(-1.0 / self.vocab_size, 1.0 / self.vocab_size)
        if self.codebook_norm:
            self.embedding.weight.data = F.normalize(self.embedding.weight.data, p=2, dim=-1)

        self.register_buffer("ema_vocab_hit_SV", torch.full((self.vocab_size,), fill_value=0.0))
        self.record_hit = 0

    def no_weight_decay(self):
        return ['embedding.weight', ]

    def forward(self, z, ret_usages=True, dropout=None):
        # one fused op instead of :750-771 + :792-799
        z_q, vq_loss, commit_loss, idx, hit_V = ops.VQStraightThrough.apply(z, self.embedding.weight, self.beta,
                                                                            self.codebook_norm)
        if ret_usages and self.training:
            # :774-788 — codebook-usage EMA (stats only, not on the gradient path)
            world = 1
            if _dist_ready():
                tdist.all_reduce(hit_V)
                world = tdist.get_world_size()
            if self.record_hit == 0:
                self.ema_vocab_hit_SV.copy_(hit_V)
            elif self.record_hit < 100:
                self.ema_vocab_hit_SV.mul_(0.9).add_(hit_V.mul(0.1))
            else:
                self.ema_vocab_hit_SV.mul_(0.99).add_(hit_V.mul(0.01))
            self.record_hit += 1
            margin = world * (z.numel() / self.z_channels) / self.vocab_size * 0.08
            codebook_usage = (self.ema_vocab_hit_SV >= margin).float().mean().item() * 100
        else:
            # the reference leaves `codebook_usage` unbound here and dies with UnboundLocalError
            # (xqgan_model.py:773-788,801); same error type, clearer message.
            raise UnboundLocalError("VectorQuantizer.forward needs ret_usages=True and train() mode "
                                    "(reference behaviour: xqgan_model.py:801); use f_to_idxBl_or_fhat for inference")
        self._last_indices = idx
        return z_q, [codebook_usage], vq_loss, commit_loss, 0.0

    def f_to_idxBl_or_fhat(self, z: torch.Tensor, to_fhat: bool, v_patch_nums=None) -> List[torch.Tensor]:
        """Inference twin (xqgan_model.py:803-833): [z_q (B,C,H,W)] or [indices (N,)]."""
        zq, idx, _, _ = ops.vq_forward_raw(z, self.embedding.weight, self.codebook_norm, ste=False, want_zq=to_fhat)
        return [zq.view(z.shape) if to_fhat else idx]
