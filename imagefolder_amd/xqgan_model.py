"""Host-side mirror of the reference's tokenizer model API over the MI355X kernels.

Mirrors (same names, ctor signatures, parameter/buffer names, return tuples) of reference
tokenizer/tokenizer_image/xqgan_model.py:
    ModelArgs :30-72 | VQModel :75-451 | VectorQuantizer :722-833 | VQ_8 / VQ_16 / VQ_models :845-851
    (CNN Encoder/Decoder live in cnn.py, the DINOv2-ViT wrappers in dino_enc/, VectorQuantizer2 in quant.py)
The quantizer arithmetic runs in libxq_ops.so (include/xq_ops.h); the encoder/decoder tensor ops go through
nn_ops.py.  This file keeps only the module state the reference keeps in Python.
"""
from dataclasses import dataclass, field
from math import sqrt
from typing import List

import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.distributed as tdist

from .lazy import LazyFloat, materialise, mean_lazy
from . import nn_ops, ops
from .cliploss import ClipLoss
from .cnn import Encoder, Decoder
from .dino_enc.dinov2 import DINOv2Encoder, DINOv2Decoder
from .dino_enc.vision_transformer import create_model
from .latent_perturbation import add_perturbation
from .quant import VectorQuantizer2


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
        # False: `usages` are Python floats, read with one synchronisation like upstream's .item() (:788); True (set by
        # train.TokenizerTrainStep): LazyFloat objects that are only waited for when somebody reads them (lazy.py)
        self.lazy_usages = False

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
            # read lazily: no device synchronisation inside the forward (the reference calls .item() here, :788)
            codebook_usage = materialise([LazyFloat((self.ema_vocab_hit_SV >= margin).float().mean() * 100)], self.lazy_usages)[0]
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


def orthogonal_cosine_loss(A, B):
    A_norm = A / A.norm(dim=1, keepdim=True)
    B_norm = B / B.norm(dim=1, keepdim=True)
    return (A_norm * B_norm).sum(dim=1).mean()


@dataclass
class ModelArgs:
    """reference xqgan_model.py:30-72 (field for field)"""
    codebook_size: int = 16384
    codebook_embed_dim: int = 8
    codebook_l2_norm: bool = True
    codebook_show_usage: bool = True
    commit_loss_beta: float = 0.25
    entropy_loss_ratio: float = 0.0

    encoder_ch_mult: List[int] = field(default_factory=lambda: [1, 1, 2, 2, 4])
    decoder_ch_mult: List[int] = field(default_factory=lambda: [1, 1, 2, 2, 4])
    z_channels: int = 256
    dropout_p: float = 0.0

    v_patch_nums: List[int] = field(default_factory=lambda: [1, 2, 3, 4, 5, 6, 8, 10, 13, 16])
    enc_type: str = 'cnn'
    dec_type: str = 'cnn'
    semantic_guide: str = 'dinov2'
    detail_guide: str = 'clip'
    num_latent_tokens: int = 256
    encoder_model: str = 'vit_small_patch14_dinov2.lvd142m'
    decoder_model: str = 'vit_small_patch14_dinov2.lvd142m'
    abs_pos_embed: bool = False
    share_quant_resi: int = 4
    product_quant: int = 1
    codebook_drop: float = 0.0
    half_sem: bool = False
    start_drop: int = 1
    sem_loss_weight: float = 0.1
    detail_loss_weight: float = 0.1
    clip_norm: bool = False
    sem_loss_scale: float = 1.0
    detail_loss_scale: float = 1.0
    guide_type_1: str = "class"
    guide_type_2: str = "class"

    lfq: bool = False
    scale: float = 1.0
    soft_entropy: bool = True

    dependency_loss_weight: float = 0.0

    test_model: bool = False


class _Affine(nn.Module):
    """datasets/normalize.py Normalize / Denormalize with the constants as (non-persistent) buffers so that they
    follow .to(device) — upstream pins them to 'cuda' at construction (datasets/normalize.py:8-13)."""

    def __init__(self, mean, std, inverse: bool):
        super().__init__()
        self.register_buffer("mean", torch.tensor(mean).view(1, -1, 1, 1), persistent=False)
        self.register_buffer("std", torch.tensor(std).view(1, -1, 1, 1), persistent=False)
        self.inverse = inverse

    def forward(self, x):
        return x * self.std + self.mean if self.inverse else (x - self.mean) / self.std


class VQModel(nn.Module):
    """Drop-in for reference VQModel (xqgan_model.py:75-451): encoder -> quant_conv -> (product) quantizer(s)
    [+ latent perturbation] -> post_quant_conv -> decoder, plus the frozen-DINOv2 semantic regulariser.
    lfq=True builds the LFQ/BSQ sign quantizer (lookup_free_quantize.py).
    Not mirrored (never enabled by the BASELINE yamls): detail_guide != 'none'
    (needs the CLIP ViT-B checkpoint)."""

    def __init__(self, config: ModelArgs):
        super().__init__()
        self.config = config
        self.enc_type = config.enc_type
        self.dec_type = config.dec_type
        self.product_quant = config.product_quant
        self.half_sem = config.half_sem
        self.start_drop = config.start_drop
        self.clip_norm = config.clip_norm
        config.num_latent_tokens = config.num_latent_tokens * config.product_quant  # :85
        vit_kwargs = {'img_size': 256, 'patch_size': 16, 'drop_path_rate': 0.1}
        vit_kwargs.update(getattr(config, "vit_overrides", None) or {})  # test hook: shrink the ViT, not upstream

        if config.enc_type == 'cnn':
            self.encoder = Encoder(ch_mult=config.encoder_ch_mult, z_channels=config.z_channels, dropout=config.dropout_p)
            self.quant_conv = nn.Conv2d(config.z_channels, config.codebook_embed_dim, 1)
        elif config.enc_type == 'dinov2':
            self.encoder = DINOv2Encoder(in_channels=3, num_latent_tokens=config.num_latent_tokens,
                                         model_name=config.encoder_model, model_kwargs=dict(vit_kwargs), pretrained=True,
                                         tuning_method='full', tuning_kwargs={'r': 8}, abs_pos_embed=config.abs_pos_embed,
                                         product_quant=config.product_quant)
            self.quant_conv = nn.Conv2d(self.encoder.embed_dim, config.codebook_embed_dim, 1)
        else:
            raise NotImplementedError

        if config.dec_type == 'cnn':
            self.decoder = Decoder(ch_mult=config.decoder_ch_mult, z_channels=config.z_channels, dropout=config.dropout_p)
            self.post_quant_conv = nn.Conv2d(config.codebook_embed_dim, config.z_channels, 1)
        elif config.dec_type == 'dinov2':
            self.decoder = DINOv2Decoder(in_channels=3, num_latent_tokens=config.num_latent_tokens // self.product_quant,
                                         model_name=config.decoder_model, model_kwargs=dict(vit_kwargs), pretrained=True,
                                         tuning_method='full', tuning_kwargs={'r': 8}, to_pixel='linear', use_rope=False,
                                         cond_latent=False, abs_pos_embed=config.abs_pos_embed)
            self.post_quant_conv = nn.Conv2d(config.codebook_embed_dim, self.decoder.embed_dim, 1)

        def make_lfq(num_latent_tokens):   # xqgan_model.py:136-144 / :157-165
            from .lookup_free_quantize import LFQ
            return LFQ(config.codebook_size, config.codebook_embed_dim, v_patch_nums=config.v_patch_nums,
                       num_latent_tokens=num_latent_tokens, share_quant_resi=config.share_quant_resi,
                       codebook_drop=config.codebook_drop, using_znorm=config.codebook_l2_norm, scale=config.scale,
                       entropy_weight=config.entropy_loss_ratio, soft_entropy=config.soft_entropy)
        self.V = self.vocab_size = config.codebook_size * self.product_quant
        self.Cvae = config.codebook_embed_dim * self.product_quant
        single_scale = len(config.v_patch_nums) == 1
        if self.product_quant > 1:
            if single_scale:
                self.quantizes = nn.ModuleList([
                    VectorQuantizer(config.codebook_size, config.codebook_embed_dim, config.commit_loss_beta,
                                    config.codebook_l2_norm) for _ in range(self.product_quant)])
            elif not config.lfq:
                self.quantizes = nn.ModuleList([
                    VectorQuantizer2(config.codebook_size, config.codebook_embed_dim, v_patch_nums=config.v_patch_nums,
                                     num_latent_tokens=config.num_latent_tokens // self.product_quant,
                                     share_quant_resi=config.share_quant_resi, codebook_drop=config.codebook_drop)
                    for _ in range(self.product_quant)])
            else:
                self.quantizes = nn.ModuleList([make_lfq(config.num_latent_tokens // self.product_quant)
                                                for _ in range(self.product_quant)])
            out_dim = self.decoder.embed_dim if config.dec_type == 'dinov2' else config.z_channels
            self.post_quant_conv = nn.Conv2d(config.codebook_embed_dim * self.product_quant, out_dim, 1)
        else:
            if single_scale:
                self.quantize = VectorQuantizer(config.codebook_size, config.codebook_embed_dim, config.commit_loss_beta,
                                                config.codebook_l2_norm)
            elif not config.lfq:
                self.quantize = VectorQuantizer2(config.codebook_size, config.codebook_embed_dim,
                                                 v_patch_nums=config.v_patch_nums,
                                                 num_latent_tokens=config.num_latent_tokens,
                                                 share_quant_resi=config.share_quant_resi)
            else:
                self.quantize = make_lfq(config.num_latent_tokens)

        self.codebook_embed_dim = config.codebook_embed_dim
        self.device_dropout_rng = False     # True: quantizer-dropout depths from the device RNG (replay-safe), see forward()
        self.v_patch_nums = config.v_patch_nums
        self.codebook_drop = config.codebook_drop
        self.semantic_guide = config.semantic_guide
        self.denormalize = _Affine([0.5, 0.5, 0.5], [0.5, 0.5, 0.5], inverse=True)
        self.normalize = _Affine([0.485, 0.456, 0.406], [0.229, 0.224, 0.225], inverse=False)
        if self.semantic_guide == 'dinov2':
            sem_kwargs = dict(img_size=256, patch_size=16, drop_path_rate=0.0)
            sem_kwargs.update(getattr(config, "vit_overrides", None) or {})
            sem_kwargs['drop_path_rate'] = 0.0
            semantic_model = create_model(config.encoder_model, pretrained=True, **sem_kwargs)
            semantic_model.eval()
            for p in semantic_model.parameters():
                p.requires_grad = False
            self.semantic_model = semantic_model
            rank = tdist.get_rank() if _dist_ready() else 0
            world_size = tdist.get_world_size() if _dist_ready() else 1
            self.sem_loss_scale = config.sem_loss_scale
            self.semantic_loss = ClipLoss(local_loss=False, gather_with_grad=True, cache_labels=True, rank=rank,
                                          world_size=world_size, use_horovod=False)
            if not self.half_sem and self.product_quant > 1:
                self.sem_linear = nn.Conv2d(self.product_quant * config.codebook_embed_dim, config.codebook_embed_dim, 1)
            elif self.half_sem and self.product_quant == 1:
                self.sem_linear = nn.Conv2d(768, config.codebook_embed_dim // 2, 1)
            if self.enc_type == 'cnn':
                self.sem_linear = torch.nn.Linear(semantic_model.embed_dim, config.codebook_embed_dim)  # upstream: 384 (ViT-S)
            self.sem_loss_weight = config.sem_loss_weight

        self.detail_guide = config.detail_guide
        if self.detail_guide != 'none':
            raise NotImplementedError("detail_guide needs the CLIP ViT-B/16 checkpoint; the yamls run detail_guide='none' "
                                      "(xqgan_train.py passes it explicitly)")
        self.guide_type_1 = config.guide_type_1
        self.guide_type_2 = config.guide_type_2
        self.dependency_loss_weight = config.dependency_loss_weight
        self.test_mode = config.test_model
        if self.test_mode:
            self.eval()
            [p.requires_grad_(False) for p in self.parameters()]

    def train(self, mode: bool = True):
        super().train(mode)
        if self.semantic_guide == 'dinov2':
            self.semantic_model.eval()  # frozen teacher stays in eval (upstream calls .eval() once, :177)
        return self

    def finetune(self, enc_tuning_method, dec_tuning_method):
        self.encoder.finetine(enc_tuning_method)
        self.decoder.finetine(dec_tuning_method)

    # ---- :241-261 -------------------------------------------------------------------------------------------
    def _tokens_to_map(self, h):
        if self.enc_type == 'dinov2':
            b, l, c = h.shape
            if self.product_quant > 1:
                assert int(sqrt(l // self.product_quant)) ** 2 * self.product_quant == l
                h = h.view(b, l, 1, c).permute(0, 3, 1, 2)
            else:
                assert int(sqrt(l)) ** 2 == l
                h = h.view(b, int(sqrt(l)), int(sqrt(l)), c).permute(0, 3, 1, 2)
        return h

    def encode(self, x):
        h = self._tokens_to_map(self.encoder(x))
        return nn_ops.conv1x1(h, self.quant_conv.weight, self.quant_conv.bias)

    def decode(self, quant, return_quant=False):
        quant = nn_ops.conv1x1(quant, self.post_quant_conv.weight, self.post_quant_conv.bias)
        if self.dec_type == 'dinov2':
            quant = quant.flatten(2).permute(0, 2, 1)
        return self.decoder(quant)

    # ---- :268-365 -------------------------------------------------------------------------------------------
    def forward(self, input, epoch, alpha, beta, delta):
        from ._lib import marker, marker_on_grad   # section boundaries for kernel traces (no-ops unless XQ_MARKERS=1)
        marker(20)
        h = self.encode(input)
        marker(21)
        marker_on_grad(h, 61)       # backward: quantizer done, encoder backward starts
        b, c, l, _ = h.shape
        if len(self.v_patch_nums) == 1:
            dropout_rand = None
        elif self.device_dropout_rng and input.is_cuda:
            # same distribution, drawn by the device generator: no host decision inside the step, so a hipGraph replay draws new
            # depths every step (train.CapturedStep switches this on; the host draw below would be frozen at capture time)
            dropout_rand = torch.randint(self.start_drop, len(self.v_patch_nums) + 1, (b,), device=input.device)
        else:  # host RNG like upstream (:274): fixes the dropout depth across the product quantizers
            dropout_rand = torch.randint(self.start_drop, len(self.v_patch_nums) + 1, (b,))

        self._last_dropout_rand = dropout_rand     # the depths this forward drew (tests / logging; device tensor on the device path)
        if self.product_quant > 1:
            side = int(sqrt(l // self.product_quant))
            quant_list, usages_list, vq_list, commit_list, ent_list = [], [], [], [], []
            for i, hi in enumerate(h.chunk(chunks=self.product_quant, dim=2)):
                hi = hi.reshape(b, -1, side, side)
                quant, usages, vq_loss, commit_loss, entropy_loss = self.quantizes[i].forward(hi, ret_usages=True,
                                                                                              dropout=dropout_rand)
                quant_list.append(quant); usages_list.append(usages); vq_list.append(vq_loss)
                commit_list.append(commit_loss); ent_list.append(entropy_loss)
            dependency_loss = self.dependency_loss_weight * orthogonal_cosine_loss(
                torch.mean(quant_list[0], dim=(2, 3)).contiguous(), torch.mean(quant_list[-1], dim=(2, 3)).contiguous())
            usages = mean_lazy(usages_list)     # :287 sum(us) / product_quant, without reading the statistics on the host
            mean_vq_loss = sum(vq_list) / self.product_quant
            mean_commit_loss = sum(commit_list) / self.product_quant
            mean_entropy = sum(ent_list) / self.product_quant
            quant = torch.cat(quant_list, dim=1)
        else:
            dependency_loss = 0.0
            quant, usages, mean_vq_loss, mean_commit_loss, mean_entropy = self.quantize.forward(h, ret_usages=True,
                                                                                                dropout=dropout_rand)
            # upstream also print()s (alpha, beta, delta) to stdout every step here (:296) — not mirrored
            if isinstance(self.quantize, VectorQuantizer):
                quant = add_perturbation(h, quant, self.quantize.z_channels, self.quantize.codebook_norm,
                                         self.quantize.embedding, alpha, beta, delta)
            else:
                raise AttributeError("upstream dereferences quantize.z_channels here (:297), which VectorQuantizer2 "
                                     "lacks: a P=1 multi-scale model cannot run its forward upstream either")
            quant_list = [quant]

        marker(22)
        marker_on_grad(quant, 60)   # backward: decoder done, quantizer backward starts
        dec = self.decode(quant)
        marker(23)

        if self.semantic_guide != 'none':
            with torch.no_grad():
                from . import ops_dense
                pe = getattr(self.semantic_model, "patch_embed", None)
                pp = pe.patch_size[0] if pe is not None else 0
                if (ops_dense.image_prep_supported(input) and pp and pp % 8 == 0 and input.shape[2] == input.shape[3]
                        and input.shape[2] % pp == 0):
                    # normalize(denormalize(x)) = x * (s1 / s2) + (m1 - m2) / s2 per channel: folded into the teacher's patchify kernel
                    # (four element-wise passes over the image batch + the permute-copy + the cast -> one kernel)
                    dn, nm = self.denormalize, self.normalize
                    key = (dn.mean._version, dn.std._version, nm.mean._version, nm.std._version, nm.std.data_ptr())
                    if getattr(self, "_sem_affine", (None,))[0] != key:
                        s1, m1 = dn.std.flatten().tolist(), dn.mean.flatten().tolist()
                        s2, m2 = nm.std.flatten().tolist(), nm.mean.flatten().tolist()
                        self._sem_affine = (key, (tuple(a / b_ for a, b_ in zip(s1, s2)), tuple((a - c_) / b_ for a, c_, b_ in zip(m1, m2, s2))))
                    aff = self._sem_affine[1]
                    if self.guide_type_1 == 'class':
                        z_s = self.semantic_model(input, input_affine=aff)[..., None, None]
                    else:
                        z_s = self.semantic_model.forward_features(input, input_affine=aff)[:, 1:, :].reshape(b, 768, 16, 16)
                else:
                    inp = self.normalize(self.denormalize(input))
                    if self.guide_type_1 == 'class':
                        z_s = self.semantic_model(inp)[..., None, None]
                    else:
                        z_s = self.semantic_model.forward_features(inp)[:, 1:, :].reshape(b, 768, 16, 16)
            if self.enc_type == 'dinov2':
                z_s = nn_ops.conv1x1(z_s, self.quant_conv.weight, self.quant_conv.bias).contiguous()
                z_s = torch.mean(z_s, dim=(2, 3)).contiguous()
                z_q_ = torch.mean(quant_list[-1], dim=(2, 3)).contiguous()
            else:
                z_q_ = torch.mean(h, dim=(2, 3)).contiguous()
                z_s = self.sem_linear(z_s.flatten(1)).contiguous()
            n_drop = int(b * self.codebook_drop)
            with torch.autocast(device_type=input.device.type, enabled=False):
                sem_loss_scale = self.sem_loss_scale
                feat1 = z_s[n_drop:].float()
                feat2 = z_q_[n_drop:].float()
                if self.clip_norm:
                    feat1 = feat1 / feat1.norm(dim=1, keepdim=True)
                    feat2 = feat2 / feat2.norm(dim=1, keepdim=True)
                    sem_loss_scale = (epoch % 200) / 200 * (100 - sem_loss_scale) + sem_loss_scale if epoch < 200 else 100
                sem_loss = self.semantic_loss.forward(feat1, feat2, logit_scale=sem_loss_scale) * self.sem_loss_weight
        else:
            sem_loss = None
        detail_loss = None
        marker(24)
        return dec, (mean_vq_loss, mean_commit_loss, mean_entropy, usages), sem_loss, detail_loss, dependency_loss

    # ---- :367-403 -------------------------------------------------------------------------------------------
    def img_to_reconstructed_img(self, x, last_one=True):
        f = nn_ops.conv1x1(self._tokens_to_map(self.encoder(x)), self.quant_conv.weight, self.quant_conv.bias)
        multi = len(self.v_patch_nums) > 1
        if self.product_quant > 1:
            b, c, l, _ = f.shape
            side = int(sqrt(l // self.product_quant))
            f_list = [fi.reshape(b, -1, side, side) for fi in f.chunk(chunks=self.product_quant, dim=2)]
            f_hats_list = [self.quantizes[i].f_to_idxBl_or_fhat(fi, to_fhat=True,
                                                                v_patch_nums=self.v_patch_nums if multi else None)
                           for i, fi in enumerate(f_list)]
            f_hats = [nn_ops.conv1x1(torch.cat(fh, dim=1), self.post_quant_conv.weight, self.post_quant_conv.bias)
                      for fh in zip(*f_hats_list)]
        else:
            ls = self.quantize.f_to_idxBl_or_fhat(f, to_fhat=True, v_patch_nums=self.v_patch_nums if multi else None)
            f_hats = [nn_ops.conv1x1(fh, self.post_quant_conv.weight, self.post_quant_conv.bias) for fh in ls]
        if self.dec_type == 'dinov2':
            f_hats = [fh.flatten(2).permute(0, 2, 1) for fh in f_hats]
        if last_one:
            return self.decoder(f_hats[-1]).clamp_(-1, 1)
        return [self.decoder(fh).clamp_(-1, 1) for fh in f_hats]

    def img_to_idx(self, x):
        """code indices of an image batch (the 'indices bit-exact' contract): list over product branches of the
        per-scale index tensors returned by f_to_idxBl_or_fhat(to_fhat=False)."""
        f = nn_ops.conv1x1(self._tokens_to_map(self.encoder(x)), self.quant_conv.weight, self.quant_conv.bias)
        multi = len(self.v_patch_nums) > 1
        vp = self.v_patch_nums if multi else None
        if self.product_quant > 1:
            b, c, l, _ = f.shape
            side = int(sqrt(l // self.product_quant))
            return [self.quantizes[i].f_to_idxBl_or_fhat(fi.reshape(b, -1, side, side), to_fhat=False, v_patch_nums=vp)
                    for i, fi in enumerate(f.chunk(chunks=self.product_quant, dim=2))]
        return [self.quantize.f_to_idxBl_or_fhat(f, to_fhat=False, v_patch_nums=vp)]

    def img_to_sem_feat(self, x):
        """quantised latent of the LAST product branch at the last scale (xqgan_model.py:405-426 upstream; the feature map the
        semantic probes read).  Upstream indexes `self.quantizes` unconditionally, which only exists for product_quant > 1; the
        single-quantizer model is served through `self.quantize` here instead of failing."""
        f = nn_ops.conv1x1(self._tokens_to_map(self.encoder(x)), self.quant_conv.weight, self.quant_conv.bias)
        multi = len(self.v_patch_nums) > 1
        vp = self.v_patch_nums if multi else None
        if self.product_quant > 1:
            b, c, l, _ = f.shape
            side = int(sqrt(l // self.product_quant))
            fi = f.chunk(chunks=self.product_quant, dim=2)[-1].reshape(b, -1, side, side)
            return self.quantizes[-1].f_to_idxBl_or_fhat(fi, to_fhat=True, v_patch_nums=vp)[-1]
        return self.quantize.f_to_idxBl_or_fhat(f, to_fhat=True, v_patch_nums=vp)[-1]

    def decode_code(self, code_b):
        """xqgan_model.py:263-266 upstream, statement for statement: written for a quantizer whose forward returns three values
        (models/quant.py); with the XQ-GAN quantizers (five values) the unpacking raises ValueError there and here."""
        quant_b, usages, mean_vq_loss = self.quantize(code_b, ret_usages=True)
        return self.decode(quant_b)

    def fhat_to_img(self, f_hat: torch.Tensor):
        f_hat = nn_ops.conv1x1(f_hat, self.post_quant_conv.weight, self.post_quant_conv.bias)
        if self.dec_type == 'dinov2':
            f_hat = f_hat.flatten(2).permute(0, 2, 1)
        return self.decoder(f_hat).clamp_(-1, 1)

    def idxBl_to_var_input(self, gt_idx_Bl):
        if self.product_quant > 1:
            return torch.cat([self.quantizes[i].idxBl_to_var_input(gt_idx_Bl[i]) for i in range(self.product_quant)], dim=-1)
        return self.quantize.idxBl_to_var_input(gt_idx_Bl)

    def get_next_autoregressive_input(self, si, SN, f_hat, h_BChw):
        outs = [self.quantizes[i].get_next_autoregressive_input(si, SN, fh, hb)
                for i, (fh, hb) in enumerate(zip(f_hat.chunk(self.product_quant, dim=1), h_BChw.chunk(self.product_quant, dim=1)))]
        return torch.cat([o[0] for o in outs], dim=1), torch.cat([o[1] for o in outs], dim=1)


def VQ_8(**kwargs):
    return VQModel(ModelArgs(encoder_ch_mult=[1, 2, 2, 4], decoder_ch_mult=[1, 2, 2, 4], **kwargs))


def VQ_16(**kwargs):
    return VQModel(ModelArgs(encoder_ch_mult=[1, 1, 2, 2, 4], decoder_ch_mult=[1, 1, 2, 2, 4], **kwargs))


VQ_models = {'VQ-16': VQ_16, 'VQ-8': VQ_8}
