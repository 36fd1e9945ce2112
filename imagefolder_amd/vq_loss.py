"""Generator / discriminator losses of the tokenizer train step — host mirror in plain tensor ops.

SURVEY.md §8f lists `VQLoss` internals as "next" after the quantizer/encoder/decoder path: they "stay PyTorch"
(library ops on the GPU), but the train step is not a train step without them, so the module is mirrored here with
the reference's structure and state-dict names:
    VQLoss        tokenizer/tokenizer_image/vq_loss.py:80-261 (disc_type='dinodisc' only — what every yaml selects)
    LPIPS         tokenizer/tokenizer_image/lpips.py:53-163   (VGG16 trunk restated: torchvision is absent offline)
    DinoDisc      tokenizer/tokenizer_image/discriminator_dino.py:26-363 (frozen DINO ViT-S/16 + 5 spectral-norm heads)
    DiffAug       tokenizer/tokenizer_image/diffaug.py:22-118
No network / checkpoints in this build: the VGG16 and DINO trunks are random-init unless a state_dict is loaded
(names match torchvision's `features.N` slices / the DINO checkpoint).  wandb logging is not mirrored.
"""
import math
import random
from typing import List

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.utils.spectral_norm import SpectralNorm


# ---- scalar loss helpers (vq_loss.py:17-78) -------------------------------------------------------------------
def hinge_d_loss(logits_real, logits_fake):
    return 0.5 * (torch.mean(F.relu(1. - logits_real)) + torch.mean(F.relu(1. + logits_fake)))


def hinge_gen_loss(logit_fake):
    return -torch.mean(logit_fake)


def adopt_weight(weight, global_step, threshold=0, value=0.):
    return value if global_step < threshold else weight


class LeCAM_EMA(object):
    """vq_loss.py:37-48 upstream: running means of the discriminator logits.  Upstream reads both means back with .item() —
    two device synchronisations in the middle of every train step; here the pair lives in a 2-element device tensor updated in
    place (no host round trip, hipGraph-capturable); on CPU tensors the host-float arithmetic of the reference is kept."""

    def __init__(self, init=0., decay=0.999):
        self.logits_real_ema = init
        self.logits_fake_ema = init
        self.decay = decay
        self._dev = None

    def update(self, logits_real, logits_fake):
        m_real, m_fake = logits_real.detach().float().mean(), logits_fake.detach().float().mean()
        if logits_real.is_cuda:
            if self._dev is None:
                self._dev = torch.tensor([float(self.logits_real_ema), float(self.logits_fake_ema)], dtype=torch.float32,
                                         device=logits_real.device)
                self.logits_real_ema, self.logits_fake_ema = self._dev[0], self._dev[1]      # views: follow the in-place updates
            self._dev.mul_(self.decay).add_(torch.stack([m_real, m_fake]), alpha=1 - self.decay)
            return
        means = torch.stack([m_real, m_fake]).tolist()
        self.logits_real_ema = self.logits_real_ema * self.decay + means[0] * (1 - self.decay)
        self.logits_fake_ema = self.logits_fake_ema * self.decay + means[1] * (1 - self.decay)


def lecam_reg(real_pred, fake_pred, lecam_ema):
    return torch.mean(F.relu(real_pred - lecam_ema.logits_fake_ema).pow(2)) + \
        torch.mean(F.relu(lecam_ema.logits_real_ema - fake_pred).pow(2))


# ---- LPIPS (lpips.py) -------------------------------------------------------------------------------------------
FUSED_VGG_BACKWARD = __import__("os").environ.get("XQ_FUSED_VGG", "1") == "1"
FUSED_DIFFAUG = __import__("os").environ.get("XQ_FUSED_DIFFAUG", "1") == "1"
FUSED_SPECTRAL_NORM = __import__("os").environ.get("XQ_FUSED_SN", "1") == "1"
# round 5: the power iterations of the five heads' same-shaped convolutions as one batched launch chain per shape (XQ_BATCHED_SN=0: per weight)
BATCHED_SPECTRAL_NORM = __import__("os").environ.get("XQ_BATCHED_SN", "1") == "1"
# round 5: ImageNet normalisation + crop / area resize + patchify + cast of the DINO-S trunk's input as one kernel per direction
FUSED_DINO_PREP = __import__("os").environ.get("XQ_FUSED_DINO_PREP", "1") == "1"
# discriminator update: reconstruction and input share one pass over the frozen DINO-S trunk (DinoDisc.forward_pair)
PAIRED_DISC_TRUNK = __import__("os").environ.get("XQ_PAIRED_DISC", "1") == "1"
# class-token readout of the discriminator trunk as one kernel per tap (ops_dense.ClsReadoutFn)
FUSED_READOUT = __import__("os").environ.get("XQ_FUSED_READOUT", "1") == "1"
_VGG16_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512]  # features[0:30]


class _VGG16Slices(nn.Module):
    """torchvision vgg16().features[0:30] cut into the five LPIPS slices (lpips.py:118-155): outputs after
    relu1_2, relu2_2, relu3_3, relu4_3, relu5_3.  Module indices equal torchvision's so `slice{k}.{idx}` keys match."""

    def __init__(self):
        super().__init__()
        layers, cin = [], 3
        for v in _VGG16_CFG:
            if v == 'M':
                layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
            else:
                layers += [nn.Conv2d(cin, v, kernel_size=3, padding=1), nn.ReLU(inplace=True)]
                cin = v
        cuts = [(0, 4), (4, 9), (9, 16), (16, 23), (23, 30)]
        for si, (a, b) in enumerate(cuts):
            seq = nn.Sequential()
            for i in range(a, b):
                seq.add_module(str(i), layers[i])
            setattr(self, f"slice{si + 1}", seq)
        for p in self.parameters():
            p.requires_grad = False

    def forward(self, x):
        # channels-last activations: MIOpen's bf16 implicit-GEMM solvers are NHWC (igemm_*_nhwc_bf16); with NCHW bf16
        # it falls back to naive_conv_* kernels on gfx950 (profiles/r01_full_step_naive_conv_stats.txt: 0.75 s/step)
        if x.is_cuda and x.shape[1] != 3:      # (the 3-channel image goes into conv1_1 planar: converting it would be two wasted copies)
            x = x.contiguous(memory_format=torch.channels_last)
        from . import nn_ops
        outs = []
        for si in range(1, 6):
            mods = list(getattr(self, f"slice{si}"))
            i = 0
            while i < len(mods):
                m = mods[i]
                if isinstance(m, nn.Conv2d):  # conv + the ReLU that always follows it, as one op (fused epilogue on the HIP path)
                    x = nn_ops.conv2d(x, m.weight, m.bias, stride=1, padding=1, relu=True)
                    i += 2
                else:  # MaxPool2d(2, 2)
                    x = nn_ops.max_pool2x2(x) if (m.kernel_size, m.stride) in ((2, 2), ((2, 2), (2, 2))) else m(x)
                    i += 1
            outs.append(x)
        return outs

    def _apply(self, fn, *a, **k):
        super()._apply(fn, *a, **k)
        for m in self.modules():
            if isinstance(m, nn.Conv2d) and m.weight.is_cuda:
                m.weight.data = m.weight.data.contiguous(memory_format=torch.channels_last)
        return self


class _NetLin(nn.Module):
    def __init__(self, chn_in, use_dropout=True):
        super().__init__()
        layers = [nn.Dropout()] if use_dropout else []
        layers += [nn.Conv2d(chn_in, 1, 1, stride=1, padding=0, bias=False)]
        self.model = nn.Sequential(*layers)


class LPIPS(nn.Module):
    def __init__(self, use_dropout=True):
        super().__init__()
        self.register_buffer('shift', torch.Tensor([-.030, -.088, -.188])[None, :, None, None])
        self.register_buffer('scale', torch.Tensor([.458, .448, .450])[None, :, None, None])
        self.chns = [64, 128, 256, 512, 512]
        self.net = _VGG16Slices()
        for k, c in enumerate(self.chns):
            setattr(self, f"lin{k}", _NetLin(c, use_dropout=use_dropout))
        for p in self.parameters():
            p.requires_grad = False

    @staticmethod
    def _unit(x, eps=1e-10):
        return x / (torch.sqrt(torch.sum(x ** 2, dim=1, keepdim=True)) + eps)

    def forward(self, input, target):
        """`input` is the data (no gradient), `target` the reconstruction — the order VQLoss calls it in (vq_loss.py:169)."""
        if target.is_cuda and not input.requires_grad:
            from . import ops_dense
            both = (torch.float32, torch.bfloat16)      # (the reconstruction arrives in the autocast dtype, the data in fp32)
            if ops_dense.image_prep_supported(target, both) and ops_dense.image_prep_supported(input, both):
                # (x - shift) / scale and autocast's cast in front of conv1_1 as one pass per image batch (ops_dense.ImageAffineBf16Fn)
                key = (self.shift._version, self.scale._version, self.scale.data_ptr())
                if getattr(self, "_affine_consts", (None,))[0] != key:
                    sc = [1.0 / float(v) for v in self.scale.detach().flatten().cpu()]
                    sh = [-float(m) * k for m, k in zip(self.shift.detach().flatten().cpu(), sc)]
                    self._affine_consts = (key, tuple(sc), tuple(sh))
                _, sc, sh = self._affine_consts
                with torch.no_grad():
                    f0 = self.net(ops_dense.ImageAffineBf16Fn.apply(input, sc, sh))
                x1 = ops_dense.ImageAffineBf16Fn.apply(target, sc, sh)
            else:
                with torch.no_grad():
                    f0 = self.net((input - self.shift) / self.scale)
                x1 = (target - self.shift) / self.scale
            lins = [getattr(self, f"lin{k}").model[-1].weight for k in range(len(self.chns))]
            if FUSED_VGG_BACKWARD and torch.is_autocast_enabled("cuda") and all(f.dtype == torch.bfloat16 for f in f0):
                # the trunk + the five level comparisons as one node with a hand-driven backward (ops_dense.LpipsVggFn)
                return ops_dense.LpipsVggFn.apply(x1, self.net, f0, lins).view(-1, 1, 1, 1)
            f1 = self.net(x1)
            val = 0
            for k in range(len(self.chns)):
                val = val + ops_dense.LpipsLevelFn.apply(f0[k], f1[k], lins[k])
            return val.view(-1, 1, 1, 1)
        f0 = self.net((input - self.shift) / self.scale)
        f1 = self.net((target - self.shift) / self.scale)
        val = 0
        for k in range(len(self.chns)):
            d = (self._unit(f0[k]) - self._unit(f1[k])) ** 2
            val = val + getattr(self, f"lin{k}").model(d).mean([2, 3], keepdim=True)
        return val


# ---- DiffAug (diffaug.py:22-118) ----------------------------------------------------------------------------------
def _shift_zero_fill(x, th, tw):
    """out[b, :, i, j] = x[b, :, i + th[b], j + tw[b]] where that lies inside the image, else 0 (one gather pass)"""
    B, C, H, W = x.shape
    ii = torch.arange(H, device=x.device).view(1, H, 1) + th.view(B, 1, 1)
    jj = torch.arange(W, device=x.device).view(1, 1, W) + tw.view(B, 1, 1)
    valid = ((ii >= 0) & (ii < H) & (jj >= 0) & (jj < W)).view(B, 1, H * W)
    lin = (ii.clamp(0, H - 1) * W + jj.clamp(0, W - 1)).view(B, 1, H * W).expand(B, C, H * W)
    return (x.reshape(B, C, H * W).gather(2, lin) * valid.to(x.dtype)).view(B, C, H, W)


class _ShiftZeroFill(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, th, tw):
        ctx.save_for_backward(th, tw)
        return _shift_zero_fill(x, th, tw)

    @staticmethod
    def backward(ctx, g):
        th, tw = ctx.saved_tensors
        return _shift_zero_fill(g.contiguous(), -th, -tw), None, None


class DiffAug(object):
    def __init__(self, prob=1.0, cutout=0.2):
        self.grids = {}
        self.prob = abs(prob)
        self.using_cutout = prob > 0
        self.cutout = cutout
        self.last_blur_radius = -1
        self.kh = self.kw = None

    def _grids(self, B, x, y, dev):
        key = (B, x, y, str(dev))
        if key not in self.grids:
            self.grids[key] = torch.meshgrid(torch.arange(B, device=dev), torch.arange(x, device=dev),
                                             torch.arange(y, device=dev), indexing='ij')
        return self.grids[key]

    def aug(self, BCHW, warmup_blur_schedule: float = 0):
        if BCHW.dtype != torch.float32:
            BCHW = BCHW.float()
        if warmup_blur_schedule > 0:  # gaussian blur warm-up (:43-62)
            C = BCHW.shape[1]
            sigma = (BCHW.shape[-2] * 0.5) ** 0.5 * warmup_blur_schedule
            r = math.floor(sigma * 3)
            if r >= 1:
                if self.last_blur_radius != r:
                    self.last_blur_radius = r
                    g = torch.arange(-r, r + 1, dtype=torch.float32, device=BCHW.device).mul_(1 / sigma).square_().neg_().exp2_()
                    g.div_(g.sum())
                    self.kh = g.view(1, 1, 2 * r + 1, 1).repeat(C, 1, 1, 1).contiguous()
                    self.kw = g.view(1, 1, 1, 2 * r + 1).repeat(C, 1, 1, 1).contiguous()
                BCHW = F.pad(BCHW, [r, r, r, r], mode='reflect')
                BCHW = F.conv2d(BCHW, self.kh, groups=C)
                BCHW = F.conv2d(BCHW, self.kw, groups=C)
        if self.prob < 1e-6:
            return BCHW
        trans, color, cut = (torch.rand(3) <= self.prob).tolist()  # host RNG like upstream (:66-67)
        B, dev = BCHW.shape[0], BCHW.device
        rand01 = torch.rand(7, B, 1, 1, device=dev) if (trans or color or cut) else None
        H, W = BCHW.shape[-2:]
        if FUSED_DIFFAUG and BCHW.is_cuda and BCHW.shape[1] == 3 and rand01 is not None:
            from .ops_dense import DiffAugFn       # the three transformations below as one op (csrc/xq_aug.hip), same draws
            geom = (round(H * 0.125), round(W * 0.125), round(H * self.cutout), round(W * self.cutout))
            return DiffAugFn.apply(BCHW, rand01, geom, (int(trans), int(color), int(self.using_cutout and cut)))
        if trans:
            dh, dw = round(H * 0.125), round(W * 0.125)
            th = rand01[0].mul(2 * dh + 1).floor().long() - dh
            tw = rand01[1].mul(2 * dw + 1).floor().long() - dw
            # upstream gathers from a zero-padded copy with clamped indices (diffaug.py:72-80): out[b,:,i,j] = x[b,:,i+th,j+tw]
            # inside the image, 0 outside — a zero-filled shift, whose exact transpose is the opposite shift
            BCHW = _ShiftZeroFill.apply(BCHW, th.view(B), tw.view(B))
        if color:
            BCHW = BCHW.add(rand01[2].unsqueeze(-1).sub(0.5))
            m = BCHW.mean(dim=1, keepdim=True)
            BCHW = BCHW.sub(m).mul(rand01[3].unsqueeze(-1).mul(2)).add_(m)
            m = BCHW.mean(dim=(1, 2, 3), keepdim=True)
            BCHW = BCHW.sub(m).mul(rand01[4].unsqueeze(-1).add(0.5)).add_(m)
        if self.using_cutout and cut:
            ch, cw = round(H * self.cutout), round(W * self.cutout)
            oh = rand01[5].mul(H + (1 - ch % 2)).floor().long()
            ow = rand01[6].mul(W + (1 - cw % 2)).floor().long()
            # upstream zeroes mask[b, clamp(i + oh - ch//2), clamp(j + ow - cw//2)] for the ch x cw grid (diffaug.py:103-113).  The box
            # always overlaps the image (0 <= oh - ch//2 + ch - 1 and oh - ch//2 <= H - 1), so the clamped index set is the
            # intersection of the box with the image: an outer product of a row and a column indicator — no index_put (which is
            # not capturable in a hipGraph and sorts its indices)
            s_h, s_w = oh.view(B, 1) - ch // 2, ow.view(B, 1) - cw // 2
            ar_h, ar_w = torch.arange(H, device=dev).view(1, H), torch.arange(W, device=dev).view(1, W)
            rows = ((ar_h >= s_h) & (ar_h <= s_h + (ch - 1))).to(BCHW.dtype)
            cols = ((ar_w >= s_w) & (ar_w <= s_w + (cw - 1))).to(BCHW.dtype)
            mask = 1 - rows.unsqueeze(2) * cols.unsqueeze(1)
            BCHW = BCHW.mul(mask.unsqueeze(1))
        return BCHW


# ---- DinoDisc (discriminator_dino.py) -----------------------------------------------------------------------------
class _SABlock(nn.Module):
    """frozen DINO ViT-S block: pre-LN, no LayerScale, tanh-GELU MLP (discriminator_dino.py:37-112)"""

    def __init__(self, dim, heads, mlp_ratio, eps):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=eps)
        self.attn = nn.Module()
        self.attn.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.attn.proj = nn.Linear(dim, dim, bias=True)
        self.heads = heads
        self.attn.num_heads = heads     # attribute names of the fused block runner (ops_dense.run_blocks)
        self.norm2 = nn.LayerNorm(dim, eps=eps)
        self.mlp = nn.Module()
        self.mlp.fc1 = nn.Linear(dim, round(dim * mlp_ratio))
        self.mlp.fc2 = nn.Linear(round(dim * mlp_ratio), dim)
        self.mlp.gelu_tanh = True

    def forward(self, x):
        B, L, C = x.shape
        from . import nn_ops
        o = nn_ops.attention_qkvpacked(self.attn.qkv(self.norm1(x)), self.heads)
        x = x + self.attn.proj(o)
        return x + self.mlp.fc2(F.gelu(self.mlp.fc1(self.norm2(x)), approximate='tanh'))


class FrozenDINOSmallNoDrop(nn.Module):
    def __init__(self, depth=12, key_depths=(2, 5, 8, 11), norm_eps=1e-6, patch_size=16, in_chans=3, embed_dim=384,
                 num_heads=6, mlp_ratio=4.):
        super().__init__()
        self.embed_dim = embed_dim
        self.img_size = 224
        self.patch_embed = nn.Module()
        self.patch_embed.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size)
        self.patch_size = patch_size
        self.patch_nums = self.img_size // patch_size
        m, s = torch.tensor((0.485, 0.456, 0.406)), torch.tensor((0.229, 0.224, 0.225))
        self.register_buffer('x_scale', (0.5 / s).reshape(1, 3, 1, 1))
        self.register_buffer('x_shift', ((0.5 - m) / s).reshape(1, 3, 1, 1))
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, self.patch_nums * self.patch_nums + 1, embed_dim))
        self.key_depths = set(d for d in key_depths if d < depth)
        self.blocks = nn.Sequential(*[_SABlock(embed_dim, num_heads, mlp_ratio, norm_eps)
                                      for _ in range(max(depth, 1 + max(self.key_depths)))])
        self.norm = nn.LayerNorm(embed_dim, eps=norm_eps)
        nn.init.trunc_normal_(self.pos_embed, std=.02)  # offline stand-in for the DINO checkpoint
        self.eval()
        [p.requires_grad_(False) for p in self.parameters()]

    def preprocess(self, x):
        """ImageNet normalisation of the [-1, 1] image + the random 224-crop / area resize (discriminator_dino.py:327-337): the part of
        the forward that draws random numbers, split off so that two batches can share one pass over the frozen blocks (`trunk`)."""
        with torch.autocast(device_type=x.device.type, enabled=False):
            x = (self.x_scale * x.float()).add_(self.x_shift)
            H, W = x.shape[-2], x.shape[-1]
            if H > self.img_size and W > self.img_size and random.random() <= 0.5:  # random 224-crop (:332-333)
                i = int(torch.randint(0, H - self.img_size + 1, (1,)).item())
                j = int(torch.randint(0, W - self.img_size + 1, (1,)).item())
                x = x[..., i:i + self.img_size, j:j + self.img_size]
            else:
                x = F.interpolate(x, size=(self.img_size, self.img_size), mode='area' if H > self.img_size else 'bicubic')
        return x

    def preprocess_patches(self, x):
        """`preprocess` + the patchify of the patch embedding in one kernel (ops_dense.DinoPrepPatchFn): the bf16 patch matrix
        (B * 196, 768) the patch-embedding GEMM reads, or None when that path does not apply (CPU, fp32 parity runs, up-scaling) — the
        caller then runs `preprocess`.  Draws the same host random numbers in the same order as `preprocess`."""
        if not (FUSED_DINO_PREP and x.is_cuda and x.dim() == 4 and x.shape[1] == 3 and torch.is_autocast_enabled("cuda")
                and torch.get_autocast_dtype("cuda") == torch.bfloat16):
            return None
        H, W = x.shape[-2], x.shape[-1]
        S = self.img_size
        if not (H > S and W > S and H < 2 * S and W < 2 * S):
            return None
        from . import nn_ops, ops_dense
        if not (nn_ops.FUSED_BLOCKS and ops_dense.GEMM_IMPL == "hip"):
            return None
        if random.random() <= 0.5:                                  # random 224-crop (:332-333)
            mode = 0
            i = int(torch.randint(0, H - S + 1, (1,)).item())
            j = int(torch.randint(0, W - S + 1, (1,)).item())
        else:
            mode, i, j = 1, 0, 0
        key = (self.x_scale._version, self.x_shift._version, self.x_scale.data_ptr())
        if getattr(self, "_prep_consts", (None,))[0] != key:       # host copies of the two normalisation constants (read once)
            self._prep_consts = (key, tuple(float(v) for v in self.x_scale.detach().flatten().cpu()),
                                 tuple(float(v) for v in self.x_shift.detach().flatten().cpu()))
        return ops_dense.DinoPrepPatchFn.apply(x, mode, i, j, S, self.patch_size, self._prep_consts[1], self._prep_consts[2])

    def forward(self, x, grad_ckpt=False) -> List[torch.Tensor]:
        cols = self.preprocess_patches(x)
        return self.trunk(self.preprocess(x)) if cols is None else self.trunk(None, cols=cols)

    def trunk(self, x, cols=None) -> List[torch.Tensor]:
        """patch embedding + the frozen blocks on a preprocessed (B, 3, 224, 224) batch — or on its patch matrix `cols` (preprocess_patches);
        every op acts per sample"""
        from . import nn_ops
        if cols is not None:
            n_tok = self.patch_nums * self.patch_nums
            x = nn_ops.linear(cols, nn_ops._weight_2d(self.patch_embed.proj.weight), self.patch_embed.proj.bias).view(cols.shape[0] // n_tok, n_tok, -1)
        else:
            x = nn_ops.patch_embed(x, self.patch_embed.proj.weight, self.patch_embed.proj.bias, self.patch_size)  # conv as GEMM
        from . import ops_dense as _od
        if x.dim() == 3 and _od.token_assemble_supported(x, x.shape[-1]):
            # [cls | patches] + position table in one kernel (ops_dense.TokenAssembleFn); the table of the frozen trunk is a constant
            key = (self.cls_token._version, self.pos_embed._version, self.pos_embed.data_ptr())
            if getattr(self, "_xq_token_table", (None,))[0] != key:
                with torch.no_grad(), torch.autocast(device_type=x.device.type, enabled=False):
                    tbl = torch.cat((self.cls_token, torch.zeros(1, x.shape[1], x.shape[2], device=x.device)), dim=1) + self.pos_embed
                self._xq_token_table = (key, tbl)
            x = _od.TokenAssembleFn.apply(x, self._xq_token_table[1], 1, False)
        else:
            with torch.autocast(device_type=x.device.type, enabled=False):
                x = torch.cat((self.cls_token.expand(x.shape[0], -1, -1), x.float()), dim=1) + self.pos_embed
        if x.is_cuda and nn_ops.FUSED_BLOCKS:
            from . import ops_dense
            blocks = list(self.blocks)
            if ops_dense.fused_supported(x, blocks):  # fused row kernels + attention kernels, as the tokenizer's ViT blocks
                act = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled("cuda") else torch.float32
                if FUSED_READOUT and ops_dense.cls_readout_supported(x):
                    # tokens + class token straight into the heads' dtype (they cast their input first thing): one kernel per tap
                    readout = lambda t: ops_dense.ClsReadoutFn.apply(t, act).transpose(1, 2)
                else:
                    readout = lambda t: (t[:, 1:] + t[:, :1]).transpose(1, 2)
                acts = [readout(x)]
                _, tapped = ops_dense.run_blocks(blocks, x, self.norm, act, taps=self.key_depths)
                for i in sorted(tapped):
                    acts.append(readout(tapped[i]))
                return acts
        with torch.autocast(device_type=x.device.type, enabled=False):
            acts = [(x[:, 1:] + x[:, :1]).transpose(1, 2)]
        for i, b in enumerate(self.blocks):
            x = b(x)
            if i in self.key_depths:
                acts.append((x[:, 1:].float() + x[:, :1].float()).transpose(1, 2))
        return acts


class _SpectralConv1d(nn.Conv1d):
    """Conv1d with spectral normalisation, state-dict compatible with torch's SpectralNorm (weight_orig, weight_u,
    weight_v; discriminator_dino.py:121-124).  Two deliberate differences in *how* it is evaluated:
      * the power iteration runs in fp32 outside autocast — under bf16 autocast torch.mv becomes a bf16 gemv that costs
        ~5 ms per call on this stack (15 convs x 2 gemv = 170 ms per discriminator forward, measured);
      * the convolution is a GEMM (kernel 1: channel matmul; kernel k, circular padding: k shifted copies gathered into
        (B*L, C_in*k) x W^T; k rolled matmuls were measured slower in the backward)."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        SpectralNorm.apply(self, name='weight', n_power_iterations=1, dim=0, eps=1e-12)
        for k, hook in list(self._forward_pre_hooks.items()):  # keep the parametrisation, drop the autocast-ed hook
            if isinstance(hook, SpectralNorm):
                del self._forward_pre_hooks[k]

    def _normalised_weight(self):
        with torch.autocast(device_type=self.weight_orig.device.type, enabled=False):
            W = self.weight_orig
            if self.training and W.is_cuda and W.dtype == torch.float32 and FUSED_SPECTRAL_NORM:
                from .ops_dense import SpectralNormWeightFn
                return SpectralNormWeightFn.apply(W, self.weight_u, self.weight_v, 1e-12)
            Wm = W.reshape(W.shape[0], -1)
            u, v = self.weight_u, self.weight_v
            if self.training:  # one power iteration per training forward, like SpectralNorm.compute_weight
                with torch.no_grad():
                    v = F.normalize(torch.mv(Wm.t(), u), dim=0, eps=1e-12, out=v)
                    u = F.normalize(torch.mv(Wm, v), dim=0, eps=1e-12, out=u)
                u, v = u.clone(), v.clone()
            sigma = torch.dot(u, torch.mv(Wm, v))
            return W / sigma

    def forward(self, x):
        W, k = self._normalised_weight(), self.kernel_size[0]
        if k == 1:
            y = torch.matmul(W[:, :, 0].to(x.dtype), x)
        else:
            pad = k // 2
            xp = F.pad(x, (pad, pad), mode='circular') if self.padding_mode == 'circular' else F.pad(x, (pad, pad))
            B, C, L = x.shape
            cols = xp.unfold(2, k, 1).permute(0, 2, 1, 3).reshape(B * L, C * k)   # (B*L, C_in*k), (c, tap) order
            y = torch.mm(cols, W.reshape(W.shape[0], -1).t().to(cols.dtype)).view(B, L, -1).permute(0, 2, 1)
        if self.bias is not None:
            y = y + self.bias.to(y.dtype)[None, :, None]
        return y


class BatchNormLocal(nn.Module):
    """batch statistics over virtual batches of 8 samples, no communication (discriminator_dino.py:127-154)"""

    def __init__(self, num_features, affine=True, virtual_bs=8, eps=1e-6):
        super().__init__()
        self.virtual_bs, self.eps, self.affine = virtual_bs, eps, affine
        if affine:
            self.weight = nn.Parameter(torch.ones(num_features))
            self.bias = nn.Parameter(torch.zeros(num_features))

    def forward(self, x):
        shape = x.size()
        x = x.float()
        G = int(np.ceil(x.size(0) / self.virtual_bs))
        x = x.view(G, -1, x.size(-2), x.size(-1))
        mean = x.mean([1, 3], keepdim=True)
        var = x.var([1, 3], keepdim=True, unbiased=False)
        x = (x - mean) / torch.sqrt(var + self.eps)
        if self.affine:
            x = x * self.weight[None, :, None] + self.bias[None, :, None]
        return x.view(shape)


class _Residual(nn.Module):
    def __init__(self, fn):
        super().__init__()
        self.fn = fn
        self.ratio = 1 / np.sqrt(2)

    def forward(self, x):
        return (self.fn(x).add(x)).mul_(self.ratio)


def _make_block(ch, ks, eps):
    return nn.Sequential(_SpectralConv1d(ch, ch, kernel_size=ks, padding=ks // 2, padding_mode='circular'),
                         BatchNormLocal(ch, eps=eps), nn.LeakyReLU(negative_slope=0.2, inplace=True))


class DinoDisc(nn.Module):
    def __init__(self, ks=9, depth=12, key_depths=(2, 5, 8, 11), norm_type='bn', using_spec_norm=True, norm_eps=1e-6):
        super().__init__()
        assert norm_type == 'bn' and using_spec_norm, "only the configuration used by VQLoss is mirrored"
        key_depths = tuple(d for d in key_depths if d < depth)
        d = FrozenDINOSmallNoDrop(depth=depth, key_depths=key_depths, norm_eps=norm_eps)
        # kept outside the module tree like upstream (a tuple, :202): invisible to parameters()/state_dict()/optimizer
        self.dino_proxy = (d,)
        C = d.embed_dim
        self.heads = nn.ModuleList([
            nn.Sequential(_make_block(C, 1, norm_eps), _Residual(_make_block(C, ks, norm_eps)),
                          _SpectralConv1d(C, 1, kernel_size=1, padding=0))
            for _ in range(len(key_depths) + 1)])

    def _apply(self, fn, *a, **k):  # the proxy must follow .to()/.cuda() even though it is not a submodule
        self.dino_proxy = (self.dino_proxy[0]._apply(fn),)
        return super()._apply(fn, *a, **k)

    def forward(self, x_in_pm1, grad_ckpt=False):
        return self._heads(self.dino_proxy[0](x_in_pm1.float()))

    def _stacked_buffer(self, convs, name, slot):
        """the H modules' `name` buffers (weight_u / weight_v) as rows of ONE tensor: the batched power iteration updates them in place.  Each
        module keeps its buffer (a view of the stack: state_dict / load_state_dict unaffected); anything that re-homes a buffer (.to(),
        a fresh registration) is noticed by its data pointer and the stack is rebuilt."""
        st = self._sn_stacks.get(slot)
        if st is None or any(c._buffers[name].data_ptr() != st[i].data_ptr() or c._buffers[name].device != st.device for i, c in enumerate(convs)):
            # (re)build: allocates and re-points the modules' buffers — never inside a stream capture (a graph would replay the power iteration
            # on the old storage), and only for fp32 buffers (the caller falls back to the per-weight path otherwise: no silent dtype change).
            # Note: the H rows of one stack alias ONE storage; state_dict() returns views of it (values and keys unchanged).
            if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
                raise RuntimeError("DinoDisc: the spectral-norm buffers were re-homed (.to() / re-registration) since the last forward; run one eager "
                                   "forward before capturing so that the stacked buffers are rebuilt outside the capture")
            assert all(c._buffers[name].dtype == torch.float32 for c in convs), "stacked spectral-norm buffers are fp32"
            st = torch.stack([c._buffers[name].detach() for c in convs]).contiguous()
            for i, c in enumerate(convs):
                c._buffers[name] = st[i]
            self._sn_stacks[slot] = st
        return st

    def _batched_normalised_weights(self):
        """[(w1, w9, wl) per head]: the 3 x H spectrally normalised weights of a training forward from three batched launch chains
        (ops_dense.spectral_norm_batched), or None when the per-weight path has to run (eval mode, CPU, the switch off)"""
        heads = list(self.heads)
        groups = [[h[0][0] for h in heads], [h[1].fn[0] for h in heads], [h[2] for h in heads]]
        c0 = groups[0][0]
        if not (FUSED_SPECTRAL_NORM and BATCHED_SPECTRAL_NORM and self.training and c0.weight_orig.is_cuda and c0.weight_orig.dtype == torch.float32
                and len(heads) > 1):
            return None
        for grp in groups:
            if any(tuple(c.weight_orig.shape) != tuple(grp[0].weight_orig.shape) or not c.training for c in grp):
                return None
            if any(c._buffers[n].dtype != torch.float32 for c in grp for n in ("weight_u", "weight_v")):
                return None      # (e.g. after .half() / .double(): the per-weight path keeps the buffers' dtype)
        from . import ops_dense
        if not hasattr(self, "_sn_stacks"):
            self._sn_stacks = {}
        per_group = []
        with torch.autocast(device_type="cuda", enabled=False):
            for gi, grp in enumerate(groups):
                u = self._stacked_buffer(grp, "weight_u", (gi, "u"))
                v = self._stacked_buffer(grp, "weight_v", (gi, "v"))
                per_group.append(ops_dense.spectral_norm_batched(grp, u, v, 1e-12))
        return [tuple(per_group[gi][hi] for gi in range(3)) for hi in range(len(heads))]

    def _heads(self, acts):
        B = acts[0].shape[0]
        from . import nn_ops
        if acts[0].is_cuda and nn_ops.FUSED_BLOCKS:
            from . import ops_dense
            if all(ops_dense.disc_head_supported(a, h) for h, a in zip(self.heads, acts)):
                nws = self._batched_normalised_weights()
                return torch.cat([ops_dense.disc_head(h, a, None if nws is None else nws[i]) for i, (h, a) in enumerate(zip(self.heads, acts))], dim=1)
        return torch.cat([h(a).view(B, -1) for h, a in zip(self.heads, acts)], dim=1)

    def forward_pair(self, make_first, make_second):
        """logits of two batches that need no gradient into the image (the discriminator update, vq_loss.py:226-261 calls the
        discriminator on the detached reconstruction, then on the input): the frozen trunk acts per sample, so both batches go
        through it in ONE pass of twice the rows — the D = 384 products of a 128-image batch fill 2.3 rounds of the chip's 256 CUs and
        are bound by their epilogues, two of them back to back 4.6 — and the heads, whose spectral-norm power iteration and virtual-batch
        statistics are per call upstream, run per batch in upstream's order.  `make_*` are called in order (augmentation draws, then
        the crop draw of that batch), so the random streams are consumed exactly as by two separate forwards."""
        d = self.dino_proxy[0]
        ia = make_first().float()
        ca = d.preprocess_patches(ia)
        xa = d.preprocess(ia) if ca is None else None
        ib = make_second().float()
        cb = d.preprocess_patches(ib)
        xb = d.preprocess(ib) if cb is None else None
        Ba = ia.shape[0]
        with torch.no_grad():
            if ca is not None and cb is not None:
                acts = d.trunk(None, cols=torch.cat([ca, cb], dim=0))
            else:                                                    # (mixed: one of the two took the unfused route — patchify it the library way)
                if xa is None or xb is None:
                    raise RuntimeError("DinoDisc.forward_pair: the two batches took different preprocessing routes")
                acts = d.trunk(torch.cat([xa, xb], dim=0))
        return self._heads([a[:Ba] for a in acts]), self._heads([a[Ba:] for a in acts])


# ---- VQLoss (vq_loss.py:80-261) -------------------------------------------------------------------------------------
class VQLoss(nn.Module):
    def __init__(self, disc_start, disc_loss="hinge", disc_dim=64, disc_type='dinodisc', image_size=256, disc_num_layers=3,
                 disc_in_channels=3, disc_weight=1.0, disc_adaptive_weight=False, gen_adv_loss='hinge',
                 reconstruction_loss='l2', reconstruction_weight=1.0, codebook_weight=1.0, perceptual_weight=1.0,
                 lecam_loss_weight=None, norm_type='bn', aug_prob=1):
        super().__init__()
        if disc_type != 'dinodisc' or disc_loss != 'hinge' or gen_adv_loss != 'hinge':
            raise NotImplementedError("only disc_type='dinodisc' with hinge losses (all reference yamls) is mirrored")
        self.disc_type = disc_type
        self.discriminator = DinoDisc(norm_type=norm_type)
        self.daug = DiffAug(prob=aug_prob, cutout=0.2)
        self.disc_loss = hinge_d_loss
        self.gen_adv_loss = hinge_gen_loss
        self.discriminator_iter_start = disc_start
        self.disc_weight = disc_weight
        self.disc_adaptive_weight = disc_adaptive_weight
        self.deposit_disc_grads_in_gen_step = False
        self.perceptual_loss = LPIPS().eval()
        self.perceptual_weight = perceptual_weight
        self.rec_loss = F.l1_loss if reconstruction_loss == "l1" else F.mse_loss
        self.rec_weight = reconstruction_weight
        self.codebook_weight = codebook_weight
        self.lecam_loss_weight = lecam_loss_weight
        if lecam_loss_weight is not None:
            self.lecam_ema = LeCAM_EMA()

    def calculate_adaptive_weight(self, nll_loss, g_loss, last_layer):
        nll_grads = torch.autograd.grad(nll_loss, last_layer, retain_graph=True)[0]
        g_grads = torch.autograd.grad(g_loss, last_layer, retain_graph=True)[0]
        d_weight = torch.norm(nll_grads) / (torch.norm(g_grads) + 1e-4)
        return torch.clamp(d_weight, 0.0, 1e4).detach()

    def _generator_loss_single_backward(self, codebook_loss, sem_loss, detail_loss, dependency_loss, inputs, reconstructions,
                                        global_step, last_layer, fade_blur_schedule):
        """Same loss value and the same gradients as the upstream generator branch (:163-196), but LPIPS/VGG and the
        discriminator are back-propagated ONCE.  Upstream runs their backward three times per step: twice inside
        calculate_adaptive_weight (two autograd.grad(..., last_layer) calls, :153-159) and again in loss.backward().
        Here d(nll)/d(recons) and d(adv)/d(recons) are computed once on a detached copy of the reconstruction, the
        last-layer gradient norms come from pushing those two cotangents through the decoder tail only, and the total
        cotangent re-enters the graph through a linear surrogate term whose VALUE equals the upstream loss."""
        from ._lib import marker
        marker(30)
        rec_leaf = reconstructions.detach().requires_grad_(True)
        with torch.enable_grad():
            rec_loss = self.rec_loss(inputs.contiguous(), rec_leaf.contiguous())
            p_loss = torch.mean(self.perceptual_loss(inputs.contiguous(), rec_leaf.contiguous()))
            null_loss = self.rec_weight * rec_loss + self.perceptual_weight * p_loss
            marker(31)
            logits_fake = self.discriminator(self.daug.aug(rec_leaf.contiguous(), fade_blur_schedule))
            generator_adv_loss = self.gen_adv_loss(logits_fake)
            marker(32)
        disc_params = [p for p in self.discriminator.parameters() if p.requires_grad] if self.deposit_disc_grads_in_gen_step else []
        g_nll = torch.autograd.grad(null_loss, rec_leaf)[0]
        marker(33)
        adv_grads = torch.autograd.grad(generator_adv_loss, [rec_leaf] + disc_params, allow_unused=True)
        marker(34)
        g_adv = adv_grads[0]
        # gradient norms w.r.t. the decoder's last layer: only the graph between last_layer and recons is traversed
        nll_ll, adv_ll = (torch.autograd.grad(reconstructions, last_layer, grad_outputs=g.to(reconstructions.dtype),
                                              retain_graph=True)[0] for g in (g_nll, g_adv))
        d_weight = torch.clamp(torch.norm(nll_ll) / (torch.norm(adv_ll) + 1e-4), 0.0, 1e4).detach()
        disc_weight = adopt_weight(self.disc_weight, global_step, threshold=self.discriminator_iter_start)
        # upstream's generator backward also deposits d(adv)/d(head params) on the discriminator; they are discarded by
        # optimizer_disc.zero_grad() (xqgan_train.py:465) before anything reads them, so they are only computed on request
        # (deposit_disc_grads_in_gen_step = True reproduces that observable state)
        for p, gp in zip(disc_params, adv_grads[1:]):
            if gp is not None:
                gp = gp * (d_weight * disc_weight)
                p.grad = gp if p.grad is None else p.grad.add_(gp)
        cot = (g_nll + (d_weight * disc_weight) * g_adv).detach()
        value = (null_loss + d_weight * disc_weight * generator_adv_loss).detach()
        surrogate = (reconstructions.float() * cot).sum()
        surrogate = surrogate + (value - surrogate.detach())  # value of the upstream loss, gradient = cot
        marker(35)
        sem_loss = 0 if sem_loss is None else sem_loss
        detail_loss = 0 if detail_loss is None else detail_loss
        dependency_loss = 0 if dependency_loss is None else dependency_loss
        return surrogate + codebook_loss[0] + codebook_loss[1] + codebook_loss[2] + sem_loss + detail_loss + dependency_loss

    def forward(self, codebook_loss, sem_loss, detail_loss, dependency_loss, inputs, reconstructions, optimizer_idx,
                global_step, last_layer=None, logger=None, log_every=100, fade_blur_schedule=0):
        if fade_blur_schedule < 1e-6:
            fade_blur_schedule = 0
        if optimizer_idx == 0:  # generator update (:163-223)
            if self.disc_adaptive_weight and reconstructions.requires_grad and last_layer is not None:
                return self._generator_loss_single_backward(codebook_loss, sem_loss, detail_loss, dependency_loss, inputs,
                                                            reconstructions, global_step, last_layer, fade_blur_schedule)
            rec_loss = self.rec_loss(inputs.contiguous(), reconstructions.contiguous())
            p_loss = torch.mean(self.perceptual_loss(inputs.contiguous(), reconstructions.contiguous()))
            logits_fake = self.discriminator(self.daug.aug(reconstructions.contiguous(), fade_blur_schedule))
            generator_adv_loss = self.gen_adv_loss(logits_fake)
            if self.disc_adaptive_weight:
                null_loss = self.rec_weight * rec_loss + self.perceptual_weight * p_loss
                disc_adaptive_weight = self.calculate_adaptive_weight(null_loss, generator_adv_loss, last_layer=last_layer)
            else:
                disc_adaptive_weight = 1
            disc_weight = adopt_weight(self.disc_weight, global_step, threshold=self.discriminator_iter_start)
            sem_loss = 0 if sem_loss is None else sem_loss
            detail_loss = 0 if detail_loss is None else detail_loss
            dependency_loss = 0 if dependency_loss is None else dependency_loss
            return self.rec_weight * rec_loss + self.perceptual_weight * p_loss + \
                disc_adaptive_weight * disc_weight * generator_adv_loss + \
                codebook_loss[0] + codebook_loss[1] + codebook_loss[2] + sem_loss + detail_loss + dependency_loss
        if optimizer_idx == 1:  # discriminator update (:226-261)
            from ._lib import marker
            marker(40)
            if PAIRED_DISC_TRUNK and hasattr(self.discriminator, "forward_pair") and reconstructions.is_cuda:
                logits_fake, logits_real = self.discriminator.forward_pair(
                    lambda: self.daug.aug(reconstructions.contiguous().detach(), fade_blur_schedule),
                    lambda: self.daug.aug(inputs.contiguous().detach(), fade_blur_schedule))
            else:
                logits_fake = self.discriminator(self.daug.aug(reconstructions.contiguous().detach(), fade_blur_schedule))
                marker(41)
                logits_real = self.discriminator(self.daug.aug(inputs.contiguous().detach(), fade_blur_schedule))
            marker(42)
            disc_weight = adopt_weight(self.disc_weight, global_step, threshold=self.discriminator_iter_start)
            if self.lecam_loss_weight is not None:
                self.lecam_ema.update(logits_real, logits_fake)
                lecam_loss = lecam_reg(logits_real, logits_fake, self.lecam_ema)
                return disc_weight * (lecam_loss * self.lecam_loss_weight + self.disc_loss(logits_real, logits_fake))
            return disc_weight * self.disc_loss(logits_real, logits_fake)
        raise ValueError(optimizer_idx)
