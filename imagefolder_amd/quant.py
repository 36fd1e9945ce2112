"""Host-side mirror of the reference multi-scale residual quantizer over the MI355X kernels.

Mirrors reference tokenizer/tokenizer_image/quant.py (class VectorQuantizer2 :13-258, Phi :261-268,
PhiShared :271-277, PhiPartiallyShared :280-291, PhiNonShared :294-305) — same constructor signature,
parameter/buffer names (`embedding.weight`, `ema_vocab_hit_SV`, `quant_resi.qresi_ls.{k}.weight|bias`) and
return tuples.  `VectorQuantizer2Var` mirrors the original VAR variant models/quant.py (3-tuple, no dropout mask).
The ladder itself (area-pool, nearest code, gather, bicubic, Phi, residual update, losses) runs in
libxq_ops.so (xq_msvq_forward / xq_msvq_backward).
"""
from math import sqrt
from typing import List, Optional, Sequence, Tuple, Union

import numpy as np
import torch
from torch import distributed as tdist, nn as nn
from torch.nn import functional as F

from .lazy import lazy_list, materialise
from . import ops


def _dist_ready() -> bool:
    return tdist.is_available() and tdist.is_initialized()


class Phi(nn.Conv2d):
    """3x3 residual conv blended with its input (quant.py:261-268).  The module form is only used by the VAR-side
    helpers; inside the training/inference ladders the same arithmetic runs in the fused HIP kernel."""

    def __init__(self, embed_dim, quant_resi):
        ks = 3
        super().__init__(in_channels=embed_dim, out_channels=embed_dim, kernel_size=ks, stride=1, padding=ks // 2)
        self.resi_ratio = abs(quant_resi)

    def forward(self, h_BChw):
        return h_BChw.mul(1 - self.resi_ratio) + super().forward(h_BChw).mul_(self.resi_ratio)


def _ticks(K):
    # quant.py:285 / :299
    return np.linspace(1 / 3 / K, 1 - 1 / 3 / K, K) if K == 4 else np.linspace(1 / 2 / K, 1 - 1 / 2 / K, K)


class PhiShared(nn.Module):
    def __init__(self, qresi: Phi):
        super().__init__()
        self.qresi: Phi = qresi

    def __getitem__(self, _) -> Phi:
        return self.qresi

    def index_of(self, _) -> int:
        return 0

    def convs(self):
        return [self.qresi]


class PhiPartiallyShared(nn.Module):
    def __init__(self, qresi_ls: nn.ModuleList):
        super().__init__()
        self.qresi_ls = qresi_ls
        K = len(qresi_ls)
        self.ticks = _ticks(K)

    def index_of(self, at_from_0_to_1: float) -> int:
        # kept in numpy exactly as upstream: si=2 of a 10-scale ladder is an exact mathematical tie
        return np.argmin(np.abs(self.ticks - at_from_0_to_1)).item()

    def __getitem__(self, at_from_0_to_1: float) -> Phi:
        return self.qresi_ls[self.index_of(at_from_0_to_1)]

    def convs(self):
        return list(self.qresi_ls)

    def extra_repr(self) -> str:
        return f'ticks={self.ticks}'


class PhiNonShared(nn.ModuleList):
    def __init__(self, qresi: List):
        super().__init__(qresi)
        K = len(qresi)
        self.ticks = _ticks(K)

    def index_of(self, at_from_0_to_1: float) -> int:
        return np.argmin(np.abs(self.ticks - at_from_0_to_1)).item()

    def __getitem__(self, at_from_0_to_1: float) -> Phi:
        return super().__getitem__(self.index_of(at_from_0_to_1))

    def convs(self):
        return list(self)

    def extra_repr(self) -> str:
        return f'ticks={self.ticks}'


class VarHelpersMixin:
    """VAR-side helpers shared by VectorQuantizer2 (quant.py:148-180,226-258 upstream) and LFQ (lookup_free_quantize.py:311-343,
    383-415): the running reconstruction f_hat built from per-scale code maps with the ladder's own kernels (ops.ms_upsample /
    ms_phi_accumulate / ms_area_pool).  Needs self.v_patch_nums, self.Cvae, self.quant_resi, self.prog_si (+ self.embedding for
    idxBl_to_var_input — LFQ has none, upstream and here: AttributeError)."""

    def _phi_at(self, si: int, SN: int):
        return self.quant_resi[si / (SN - 1)] if SN > 1 else self.quant_resi[0]

    def embed_to_fhat(self, ms_h_BChw: List[torch.Tensor], all_to_max_scale=True, last_one=False):
        """cumulative reconstructions from per-scale code embeddings (quant.py:148-180, all_to_max_scale branch)"""
        if not all_to_max_scale:
            raise NotImplementedError("experimental upstream branch (quant.py:165-178) is not mirrored")
        SN = len(self.v_patch_nums)
        HW = self.v_patch_nums[-1]
        B = ms_h_BChw[0].shape[0]
        f_hat = torch.zeros(B, self.Cvae, HW, HW, dtype=torch.float32, device=ms_h_BChw[0].device)
        outs = []
        for si in range(SN):
            u = ops.ms_upsample(ms_h_BChw[si], HW, HW, bicubic=si < SN - 1)
            ops.ms_phi_accumulate(f_hat, u, self._phi_at(si, SN))
            if not last_one:
                outs.append(f_hat.clone())
        return f_hat if last_one else outs

    def idxBl_to_var_input(self, gt_ms_idx_Bl: List[torch.Tensor]) -> torch.Tensor:
        """teacher-forcing input of VAR: the area-pooled running reconstruction before every scale but the first
        (quant.py:226-245) -> (B, sum_{s >= 1} pn_s^2, C)"""
        SN = len(self.v_patch_nums)
        HW = self.v_patch_nums[-1]
        B = gt_ms_idx_Bl[0].shape[0]
        f_hat = torch.zeros(B, self.Cvae, HW, HW, dtype=torch.float32, device=gt_ms_idx_Bl[0].device)
        nxt = []
        for si in range(SN - 1):
            if self.prog_si == 0 or (0 <= self.prog_si - 1 < si):
                break
            u = ops.ms_upsample(gt_ms_idx_Bl[si], HW, HW, codebook=self.embedding.weight, bicubic=True)
            ops.ms_phi_accumulate(f_hat, u, self._phi_at(si, SN))
            pn = self.v_patch_nums[si + 1]
            nxt.append(ops.ms_area_pool(f_hat, pn).view(B, self.Cvae, pn * pn).transpose(1, 2))
        return torch.cat(nxt, dim=1) if nxt else None

    def get_next_autoregressive_input(self, si: int, SN: int, f_hat: torch.Tensor, h_BChw: torch.Tensor):
        """one step of VAR sampling (quant.py:248-258): folds scale si into f_hat IN PLACE, returns (f_hat, next input map)"""
        HW = self.v_patch_nums[-1]
        u = ops.ms_upsample(h_BChw, HW, HW, bicubic=si != SN - 1)
        ops.ms_phi_accumulate(f_hat, u, self._phi_at(si, SN))
        if si != SN - 1:
            return f_hat, ops.ms_area_pool(f_hat, self.v_patch_nums[si + 1])
        return f_hat, f_hat


class VectorQuantizer2(VarHelpersMixin, nn.Module):
    """Drop-in for reference VectorQuantizer2 (tokenizer_image/quant.py:13-258)."""

    def __init__(self, vocab_size, Cvae, using_znorm=True, beta: float = 0.25, default_qresi_counts=0, v_patch_nums=None,
                 quant_resi=0.5, share_quant_resi=4, num_latent_tokens=256, codebook_drop=0.0):
        super().__init__()
        self.vocab_size: int = vocab_size
        self.Cvae: int = Cvae
        self.using_znorm: bool = using_znorm
        self.v_patch_nums: Tuple[int] = v_patch_nums
        self.num_latent_tokens = num_latent_tokens

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
        self.lazy_usages = False     # as VectorQuantizer.lazy_usages: Python floats at the API seam unless the train step opts in

        self.beta: float = beta
        self.embedding = nn.Embedding(self.vocab_size, self.Cvae)
        self.codebook_drop = codebook_drop

        self.embedding.weight.data.uniform_(-1.0 / self.vocab_size, 1.0 / self.vocab_size)
        if self.using_znorm:
            self.embedding.weight.data = F.normalize(self.embedding.weight.data, p=2, dim=-1)
        self.prog_si = -1  # progressive training: not supported upstream either

    def eini(self, eini):
        if eini > 0:
            nn.init.trunc_normal_(self.embedding.weight.data, std=eini)
        elif eini < 0:
            self.embedding.weight.data.uniform_(-abs(eini) / self.vocab_size, abs(eini) / self.vocab_size)

    def extra_repr(self) -> str:
        return f'{self.v_patch_nums}, znorm={self.using_znorm}, beta={self.beta}  |  S={len(self.v_patch_nums)}, quant_resi={self.quant_resi_ratio}'

    # ---- ladder configuration shared by forward / inference --------------------------------------------------
    def _phi_pack(self, patch_nums=None):
        """(phi_sel per scale, stacked weights (K,C,C,3,3), stacked biases (K,C)) or (.., None, None) for Identity."""
        SN = len(patch_nums if patch_nums is not None else self.v_patch_nums)
        convs = self.quant_resi.convs()
        if not isinstance(convs[0], Phi):
            return [0] * SN, None, None
        if SN == 1:
            sel = [self.quant_resi.index_of(0)] if not isinstance(self.quant_resi, PhiShared) else [0]
            # quant.py:110-111: SN == 1 uses quant_resi[0] (ModuleList index 0 / ticks lookup of 0)
            if isinstance(self.quant_resi, PhiNonShared):
                sel = [0]
        else:
            sel = [self.quant_resi.index_of(si / (SN - 1)) for si in range(SN)]
        w = torch.stack([c.weight for c in convs], 0)
        b = torch.stack([c.bias for c in convs], 0)
        return sel, w, b

    # ===================== `forward` is only used in VAE training =====================
    def forward(self, f_BChw: torch.Tensor, ret_usages=False, dropout=None):
        f = f_BChw if f_BChw.dtype == torch.float32 else f_BChw.float()  # quant.py:65-66
        B, C, H, W = f.shape
        SN = len(self.v_patch_nums)
        # quantizer dropout: per-sample number of active scales, built on the host exactly like quant.py:79-86
        inv_ratio = None
        if self.training and dropout is not None and dropout.is_cuda:
            # depths drawn on the device (VQModel.device_dropout_rng): everything stays there — no host read, replay-safe
            n_quantizers = torch.full((B,), float(SN + 1), device=dropout.device)
            n_dropout = int(B * self.codebook_drop)
            n_quantizers[:n_dropout] = dropout[:n_dropout].to(n_quantizers.dtype)
            active = torch.arange(SN, device=dropout.device, dtype=torch.float32)[:, None] < n_quantizers[None, :]
            inv_ratio = float(B) / active.float().sum(dim=1)            # 1 / ratio_s, ratio_s = mask.sum() / B (quant.py:128)
        elif self.training and dropout is not None:
            n_quantizers = torch.ones((B,)) * (SN + 1)
            n_dropout = int(B * self.codebook_drop)
            n_quantizers[:n_dropout] = dropout[:n_dropout].to(n_quantizers.dtype)
        else:
            n_quantizers = torch.ones((B,)) * (SN + 1)
        if inv_ratio is None:
            # ratio_s = mask.sum()/B (quant.py:128) — known on the host, no device sync
            ratio = [float((torch.full((B,), float(si)) < n_quantizers).sum().item()) / B for si in range(SN)]
        last_pn = self.v_patch_nums[-1]
        skip_last_pool = (last_pn == int(sqrt(self.num_latent_tokens)))  # quant.py:91-92
        sel, phi_w, phi_b = self._phi_pack()
        cfg = dict(patch_nums=list(self.v_patch_nums), phi_sel=sel, phi_ratio=abs(self.quant_resi_ratio),
                   using_znorm=self.using_znorm, skip_last_pool=skip_last_pool)
        f_hat, sq_vq, sq_commit, idx_all, hit_SV = ops.MSVQLadder.apply(f, self.embedding.weight, phi_w, phi_b,
                                                                        n_quantizers.to(f.device), cfg)
        numel = float(f.numel())
        if inv_ratio is None:
            inv_ratio = torch.tensor([1.0 / r for r in ratio], dtype=torch.float32, device=f.device)
        mean_vq_loss = (sq_vq * inv_ratio).sum() * (1.0 / numel / SN)          # :131,:134
        mean_commit_loss = (sq_commit * inv_ratio).sum() * (self.beta / numel)  # :132
        if self.training:
            # ONE batched (SN, V) all-reduce instead of SN async ones (quant.py:102-104,119-120); the EMA recursion
            # below is the reference's, including record_hit advancing once per SCALE (:121-127)
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
        margin = world * (f.numel() / f.shape[1]) / self.vocab_size * 0.08
        if ret_usages:
            # one ASYNCHRONOUS device->host copy for all scales, waited for when read (the reference does SN .item() syncs, quant.py:140)
            usages = materialise(lazy_list((self.ema_vocab_hit_SV >= margin).float().mean(dim=1) * 100), self.lazy_usages)
        else:
            usages = None
        self._last_indices = idx_all
        return f_hat, usages, mean_vq_loss, mean_commit_loss, 0

    # ===================== inference ladder =====================
    def f_to_idxBl_or_fhat(self, f_BChw: torch.Tensor, to_fhat: bool,
                           v_patch_nums: Optional[Sequence[Union[int, Tuple[int, int]]]] = None) -> List[torch.Tensor]:
        B, C, H, W = f_BChw.shape
        pns = [pn if isinstance(pn, int) else pn[0] for pn in (v_patch_nums or self.v_patch_nums)]
        for pn in (v_patch_nums or self.v_patch_nums):
            if not isinstance(pn, int) and pn[0] != pn[1]:
                raise ops.XqError("non-square patch sizes are not supported by the HIP ladder")
        sel, phi_w, phi_b = self._phi_pack(pns)
        r = ops.msvq_forward_raw(f_BChw, self.embedding.weight, pns, sel, phi_w, phi_b, abs(self.quant_resi_ratio),
                                 self.using_znorm, None, skip_last_pool=(pns[-1] == 16),  # hard-coded 16 upstream (:201)
                                 want_ste=False, want_saved=False, want_sq=False, want_hist=False, want_scales=to_fhat)
        if to_fhat:
            return [r["f_hat_scales"][si] for si in range(len(pns))]
        out, off = [], 0
        for pn in pns:
            n = B * pn * pn
            out.append(r["idx_all"][off:off + n].view(B, pn * pn))
            off += n
        return out

    # ===================== VAR-side helpers on the ladder primitives of libxq_ops.so =====================
    # Inference-time plumbing between the tokenizer and the VAR generator.  Each is a short sequence of the three kernels the
    # fused ladder is made of — ops.ms_upsample (code gather + bicubic), ops.ms_phi_accumulate (f_hat += Phi_k(.)) and
    # ops.ms_area_pool — so they share its arithmetic (bit-identical to oracle/xq_oracle.c); no autograd, as upstream uses them.


class VectorQuantizer2Var(VectorQuantizer2):
    """The original VAR quantizer (reference models/quant.py:52-105): no quantizer dropout, the fused loss
    mean_vq_loss = (1/SN) sum_s [beta*mse(sg f_hat_s, f) + mse(f_hat_s, sg f)] (:95-97), 3-tuple return, usage EMA
    all-reduced only when a process group exists (:79,88)."""

    def __init__(self, vocab_size, Cvae, using_znorm, beta: float = 0.25, default_qresi_counts=0, v_patch_nums=None,
                 quant_resi=0.5, share_quant_resi=4):
        super().__init__(vocab_size, Cvae, using_znorm=using_znorm, beta=beta, default_qresi_counts=default_qresi_counts,
                         v_patch_nums=v_patch_nums, quant_resi=quant_resi, share_quant_resi=share_quant_resi,
                         num_latent_tokens=int(v_patch_nums[-1]) ** 2, codebook_drop=0.0)

    def forward(self, f_BChw: torch.Tensor, ret_usages=False):
        f = f_BChw if f_BChw.dtype == torch.float32 else f_BChw.float()
        SN = len(self.v_patch_nums)
        sel, phi_w, phi_b = self._phi_pack()
        cfg = dict(patch_nums=list(self.v_patch_nums), phi_sel=sel, phi_ratio=abs(self.quant_resi_ratio),
                   using_znorm=self.using_znorm, skip_last_pool=True)  # models/quant.py:68: last scale never pooled
        f_hat, sq_vq, sq_commit, idx_all, hit_SV = ops.MSVQLadder.apply(f, self.embedding.weight, phi_w, phi_b, None, cfg)
        numel = float(f.numel())
        mean_vq_loss = (sq_commit.sum() * self.beta + sq_vq.sum()) * (1.0 / numel / SN)
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
        B, C, H, W = f.shape
        world = tdist.get_world_size() if _dist_ready() else 1
        margin = world * (B * H * W) / self.vocab_size * 0.08
        usages = materialise(lazy_list((self.ema_vocab_hit_SV >= margin).float().mean(dim=1) * 100), self.lazy_usages) if ret_usages else None
        self._last_indices = idx_all
        return f_hat, usages, mean_vq_loss
