"""Host-side mirror of the reference lookup-free (binary / sign) multi-scale quantizer over the MI355X kernels.

Mirrors reference tokenizer/tokenizer_image/lookup_free_quantize.py (class LFQ :83-415; used by the MSBR yamls through
xqgan_model.py:136-165 with `lfq: True`): same constructor signature, buffers (`ema_vocab_hit_SV`, non-persistent `mask`
and `codebook`, `scaler`), parameter names (`quant_resi.qresi_ls.{k}.weight|bias`) and return tuple
(f_hat, usages, mean_vq_loss, mean_commit_loss, mean_entropy_loss).

The residual ladder is the one of VectorQuantizer2 (area-pool -> code -> bicubic up -> Phi -> residual update, masked by the
quantizer-dropout depth) with the nearest-code search replaced by the sign pattern of the pooled residual and a FIXED
codebook of +-scaler corners: it runs in libxq_ops.so (xq_msvq_forward with using_znorm = 2 / xq_msvq_backward) on the
channel axis zero-padded to the kernels' 16 channels.  The entropy terms (:283-300, analytical per-bit form) are a few
element-wise tensor ops on f - f_hat_{<s}; they are evaluated with ordinary autograd ops.

Upstream behaviours kept on purpose: the per-sample dropout mask is used as an INTEGER index into the batch in
soft_entropy_loss (`z[mask]` with mask in {0, 1}, :285-286), so active samples read sample 1 and dropped ones sample 0;
eval-mode forward raises TypeError (`self.v_patch_nums + 1` on a tuple/list, :174).
"""
from math import sqrt
from typing import List, Optional, Sequence, Tuple, Union

import torch
from torch import distributed as tdist, nn as nn
from torch.nn import functional as F

from . import ops
from .quant import Phi, PhiNonShared, PhiPartiallyShared, PhiShared, VarHelpersMixin, _dist_ready

_CPAD = 16  # channel count of the ladder kernels the bit channels are padded to


class LFQ(VarHelpersMixin, nn.Module):
    def __init__(self, codebook_size, Cvae, using_znorm=False, beta: float = 0.25, default_qresi_counts=0, v_patch_nums=None,
                 quant_resi=0.5, share_quant_resi=4, num_latent_tokens=256, codebook_drop=0.0, scale=1,
                 sample_minimization_weight=1.0, batch_maximization_weight=1.0, entropy_weight=0.1, soft_entropy=True):
        super().__init__()
        self.Cvae: int = Cvae
        self.vocab_size: int = 2 ** self.Cvae
        assert self.vocab_size == codebook_size
        if Cvae > _CPAD:
            raise ops.XqError(f"LFQ: at most {_CPAD} bit channels are supported by the HIP ladder (got {Cvae})")
        self.using_znorm: bool = using_znorm
        self.v_patch_nums: Tuple[int] = v_patch_nums
        self.num_latent_tokens = num_latent_tokens
        self.entropy_weight = entropy_weight
        self.soft_entropy = soft_entropy
        self.persample_entropy_compute = 'analytical'

        self.quant_resi_ratio = quant_resi
        mk = lambda: (Phi(Cvae, quant_resi) if abs(quant_resi) > 1e-6 else nn.Identity())
        if share_quant_resi == 0:
            self.quant_resi = PhiNonShared([mk() for _ in range(default_qresi_counts or len(self.v_patch_nums))])
        elif share_quant_resi == 1:
            self.quant_resi = PhiShared(mk())
        else:
            self.quant_resi = PhiPartiallyShared(nn.ModuleList([mk() for _ in range(share_quant_resi)]))

        self.register_buffer('ema_vocab_hit_SV', torch.full((len(self.v_patch_nums), self.vocab_size), fill_value=0.0))
        self.record_hit = 0
        self.register_buffer('mask', 2 ** torch.arange(self.Cvae), persistent=False)
        self.beta: float = beta
        self.codebook_drop = codebook_drop

        scaler = scale ** torch.arange(len(self.v_patch_nums))
        if using_znorm:
            scaler = scaler / sqrt(self.Cvae)
        self.register_buffer('scaler', scaler)
        self.sample_minimization_weight = sample_minimization_weight
        self.batch_maximization_weight = batch_maximization_weight

        bits = self.indices_to_bits(torch.arange(codebook_size))
        self.register_buffer('codebook', bits * 2.0 - 1.0, persistent=False)
        self.prog_si = -1

    def extra_repr(self) -> str:
        return f'{self.v_patch_nums}, znorm={self.using_znorm}, beta={self.beta}  |  S={len(self.v_patch_nums)}, quant_resi={self.quant_resi_ratio}'

    # ---- :254-281 ---------------------------------------------------------------------------------------------------------
    def bits_to_indices(self, bits):
        assert bits.shape[-1] == self.Cvae
        indices = 2 ** torch.arange(0, self.Cvae, 1, dtype=torch.long, device=bits.device)
        return (bits * indices).sum(-1)

    def indices_to_bits(self, x, si=None):
        mask = 2 ** torch.arange(self.Cvae, device=x.device, dtype=torch.long)
        x = (x.unsqueeze(-1) & mask) != 0
        if si is None:
            return x
        return torch.where(x, self.scaler[si], -self.scaler[si])

    # ---- :283-310 ---------------------------------------------------------------------------------------------------------
    def get_entropy(self, count, dim=-1, eps=1e-4, normalize=True):
        if normalize:
            probs = (count + eps) / (count + eps).sum(dim=dim, keepdim=True)
        else:
            probs = count
        return -(probs * torch.log(probs + 1e-8)).sum(dim=dim)

    def soft_entropy_loss(self, z, si, codebook, mask=None):
        if mask is not None:
            z = z[mask]  # upstream: an INTEGER {0,1} index into the batch (:285-286), not a boolean selection
        if self.persample_entropy_compute != 'analytical':
            raise NotImplementedError("only the analytical per-sample entropy (the upstream default) is mirrored")
        # the (tokens x V) softmax upstream also evaluates here (:287-288) is dead code in the analytical branch: skipped
        p = torch.sigmoid(-4 * z * (self.scaler[si]))
        prob = torch.stack([p, 1 - p], dim=-1)
        per_sample_entropy = self.get_entropy(prob, dim=-1, normalize=False).sum(dim=-1).mean()
        avg_prob = prob.reshape(-1, prob.shape[-2], prob.shape[-1]).mean(0)   # reduce(prob, '... g d -> g d', 'mean')
        codebook_entropy = self.get_entropy(avg_prob, dim=-1, normalize=False)
        return per_sample_entropy, codebook_entropy.sum(), avg_prob

    # ---- ladder plumbing --------------------------------------------------------------------------------------------------
    def _phi_pack(self, SN):
        """phi_sel per scale + Phi weights/biases zero-padded to the 16-channel ladder (padded channels stay exactly 0)."""
        convs = self.quant_resi.convs()
        if not isinstance(convs[0], Phi):
            return [0] * SN, None, None
        if SN == 1:
            raise ops.XqError("LFQ with a single scale divides by SN - 1 = 0 upstream (:192)")
        sel = [self.quant_resi.index_of(si / (SN - 1)) for si in range(SN)]
        C = self.Cvae
        w = torch.stack([F.pad(c.weight, (0, 0, 0, 0, 0, _CPAD - C, 0, _CPAD - C)) for c in convs], 0)
        b = torch.stack([F.pad(c.bias, (0, _CPAD - C)) for c in convs], 0)
        return sel, w, b

    def _padded_codebook(self, dev):
        s = float(self.scaler[0])
        if not bool((self.scaler == self.scaler[0]).all()):
            raise ops.XqError("LFQ: per-scale codebook values (scale != 1) are not supported by the HIP ladder")
        return F.pad(self.codebook.to(dev).float() * s, (0, _CPAD - self.Cvae)).contiguous()

    # ===================== `forward` is only used in VAE training (:149-250) =====================
    def forward(self, f_BChw: torch.Tensor, ret_usages=False, dropout=None):
        dtype = f_BChw.dtype
        if dtype != torch.float32:
            f_BChw = f_BChw.float()
        B, C, H, W = f_BChw.shape
        if self.using_znorm:
            f_BChw = F.normalize(f_BChw, dim=1)
        SN = len(self.v_patch_nums)
        if self.training:
            n_quantizers = torch.ones((B,)) * (SN + 1)
            n_dropout = int(B * self.codebook_drop)
            n_quantizers[:n_dropout] = dropout[:n_dropout]
        else:
            n_quantizers = torch.ones((B,)) * (self.v_patch_nums + 1)  # upstream :174 — raises TypeError, kept
        masks = [(torch.full((B,), float(si)) < n_quantizers) for si in range(SN)]
        ratio = [float(m.sum().item()) / B for m in masks]
        dev = f_BChw.device
        skip_last_pool = (self.v_patch_nums[-1] == int(sqrt(self.num_latent_tokens)))   # :181
        sel, phi_w, phi_b = self._phi_pack(SN)
        cfg = dict(patch_nums=list(self.v_patch_nums), phi_sel=sel, phi_ratio=abs(self.quant_resi_ratio), using_znorm=2,
                   skip_last_pool=skip_last_pool, return_h=True)
        with torch.autocast(device_type=dev.type, enabled=False):
            f_pad = F.pad(f_BChw, (0, 0, 0, 0, 0, _CPAD - C))
            f_hat_pad, sq_vq, sq_commit, idx_all, hit_SV, h_scales = ops.MSVQLadder.apply(
                f_pad, self._padded_codebook(dev), phi_w, phi_b, n_quantizers.to(dev), cfg)
            f_hat = f_hat_pad[:, :C]
            numel = float(f_BChw.numel())
            inv_ratio = torch.tensor([1.0 / r for r in ratio], dtype=torch.float32, device=dev)
            mean_vq_loss = (sq_vq * inv_ratio).sum() * (1.0 / numel / SN)              # :234,:243
            mean_commit_loss = (sq_commit * inv_ratio).sum() * (self.beta / numel / SN)  # :235,:244

            # entropy terms on x_s = f - sg(f_hat before scale s) (:199, :221-232)
            mean_entropy_loss = 0.0
            f_hat_prev = torch.zeros_like(f_BChw)
            codebook = self.codebook.to(dev)
            for si in range(SN):
                x = (f_BChw - f_hat_prev).permute(0, 2, 3, 1).reshape(B, H * W, 1, C)
                m_int = masks[si].to(dev).int()
                if self.soft_entropy:
                    per_sample_entropy, codebook_entropy, _ = self.soft_entropy_loss(x, si, codebook * self.scaler[si], m_int)
                    entropy_aux_loss = ((self.sample_minimization_weight * per_sample_entropy)
                                        - (self.batch_maximization_weight * codebook_entropy))
                else:
                    raise NotImplementedError("soft_entropy=False (entropy over the 2^C codes, :226-232) is not mirrored")
                mean_entropy_loss = mean_entropy_loss + entropy_aux_loss * (self.entropy_weight / ratio[si])
                f_hat_prev = f_hat_prev + h_scales[si][:, :C] * masks[si].to(dev).float()[:, None, None, None]
            mean_entropy_loss = mean_entropy_loss * (1. / SN)

            if self.training:
                if _dist_ready():
                    tdist.all_reduce(hit_SV)
                for si in range(SN):
                    if self.record_hit == 0:
                        self.ema_vocab_hit_SV[si].copy_(hit_SV[si])
                    elif self.record_hit < 100:
                        self.ema_vocab_hit_SV[si].mul_(0.9).add_(hit_SV[si].mul(0.1))
                    else:
                        self.ema_vocab_hit_SV[si].mul_(0.99).add_(hit_SV[si].mul(0.01))
                    self.record_hit += 1
        world = tdist.get_world_size() if _dist_ready() else 1
        margin = world * (f_BChw.numel() / f_BChw.shape[1]) / self.vocab_size * 0.08
        if ret_usages:
            usages = ((self.ema_vocab_hit_SV >= margin).float().mean(dim=1) * 100).tolist()
        else:
            usages = None
        self._last_indices = idx_all
        return f_hat, usages, mean_vq_loss, mean_commit_loss, mean_entropy_loss

    # ===================== inference ladder (:344-381) =====================
    def f_to_idxBl_or_fhat(self, f_BChw: torch.Tensor, to_fhat: bool,
                           v_patch_nums: Optional[Sequence[Union[int, Tuple[int, int]]]] = None) -> List[torch.Tensor]:
        B, C, H, W = f_BChw.shape
        if self.using_znorm:
            f_BChw = F.normalize(f_BChw, dim=1)
        pns = [pn if isinstance(pn, int) else pn[0] for pn in (v_patch_nums or self.v_patch_nums)]
        for pn in (v_patch_nums or self.v_patch_nums):
            if not isinstance(pn, int) and pn[0] != pn[1]:
                raise ops.XqError("non-square patch sizes are not supported by the HIP ladder")
        sel, phi_w, phi_b = self._phi_pack(len(pns))
        f_pad = F.pad(f_BChw.float(), (0, 0, 0, 0, 0, _CPAD - C))
        r = ops.msvq_forward_raw(f_pad, self._padded_codebook(f_BChw.device), pns, sel, phi_w, phi_b, abs(self.quant_resi_ratio), 2,
                                 None, skip_last_pool=(pns[-1] == 16),  # hard-coded 16 upstream (:366)
                                 want_ste=False, want_saved=False, want_sq=False, want_hist=False, want_scales=to_fhat)
        if to_fhat:
            return [r["f_hat_scales"][si][:, :C] for si in range(len(pns))]
        out, off = [], 0
        for pn in pns:
            n = B * pn * pn
            out.append(r["idx_all"][off:off + n].view(B, pn * pn))
            off += n
        return out
