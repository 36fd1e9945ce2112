"""ViT backbone of the tokenizer's encoder/decoder (DINOv2 ViT-S/B/L geometry), MI355X path.

Mirrors the parts of the reference's vendored timm file that the hot path executes
(tokenizer/tokenizer_image/dino_enc/vision_transformer.py: Attention :145-197, LayerScale :280-292, Block :295-339,
VisionTransformer.__init__/_pos_embed/forward_features :587-961, vit_*_patch14_dinov2 :2886-2931) plus the timm 1.0.9
layers it imports (PatchEmbed, Mlp, DropPath, resample_abs_pos_embed; timm is not in /root/reference — the published
behaviour of those layers is restated here).  Parameter names equal timm's, so reference checkpoints
(`encoder.model.blocks.N.attn.qkv.weight`, ...) load unchanged.  The ~3000 lines of model-zoo configs are out of scope.

All tensor math goes through imagefolder_amd.nn_ops (hand-written HIP kernels where they exist).
"""
import math
from functools import partial
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import nn_ops


def trunc_normal_(t, std=.02):
    return nn.init.trunc_normal_(t, mean=0.0, std=std, a=-2.0, b=2.0)


def resample_abs_pos_embed(posemb, new_size, old_size=None, num_prefix_tokens: int = 1, interpolation: str = 'bicubic',
                           antialias: bool = True):
    """timm.layers.resample_abs_pos_embed (1.0.9): bicubic(+antialias) resize of the grid part of a position table."""
    num_pos_tokens = posemb.shape[1]
    num_new_tokens = new_size[0] * new_size[1] + num_prefix_tokens
    if num_new_tokens == num_pos_tokens and new_size[0] == new_size[1]:
        return posemb
    if old_size is None:
        hw = int(math.sqrt(num_pos_tokens - num_prefix_tokens))
        old_size = hw, hw
    if num_prefix_tokens:
        posemb_prefix, posemb = posemb[:, :num_prefix_tokens], posemb[:, num_prefix_tokens:]
    else:
        posemb_prefix = None
    embed_dim = posemb.shape[-1]
    orig_dtype = posemb.dtype
    posemb = posemb.float().reshape(1, old_size[0], old_size[1], -1).permute(0, 3, 1, 2)
    posemb = F.interpolate(posemb, size=new_size, mode=interpolation, antialias=antialias)
    posemb = posemb.permute(0, 2, 3, 1).reshape(1, -1, embed_dim).to(orig_dtype)
    if posemb_prefix is not None:
        posemb = torch.cat([posemb_prefix, posemb], dim=1)
    return posemb


class PatchEmbed(nn.Module):
    """timm PatchEmbed: non-overlapping conv (kernel = stride = patch) then NCHW -> NLC."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, bias=True):
        super().__init__()
        self.img_size = (img_size, img_size)
        self.patch_size = (patch_size, patch_size)
        self.grid_size = (img_size // patch_size, img_size // patch_size)
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size, bias=bias)
        self.norm = nn.Identity()

    def forward(self, x, affine=None):
        """affine = (scale3, shift3): a per-channel input normalisation folded into the patchify kernel (nn_ops.patch_embed)"""
        return nn_ops.patch_embed(x, self.proj.weight, self.proj.bias, self.patch_size[0], affine=affine)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, drop=0.):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features)
        self.act = act_layer()
        self.drop1 = nn.Dropout(drop)
        self.norm = nn.Identity()
        self.fc2 = nn.Linear(hidden_features, out_features)
        self.drop2 = nn.Dropout(drop)

    def forward(self, x):
        x = nn_ops.linear_gelu(x, self.fc1.weight, self.fc1.bias)  # fc1 + exact (erf) GELU
        return nn_ops.linear(x, self.fc2.weight, self.fc2.bias)


class DropPath(nn.Module):
    """Stochastic depth per sample (timm DropPath, scale_by_keep=True)."""

    # parity tests (SURVEY §7 "pass RNG draws as tensors"): when REPLAY is a list, keep_mask pops the recorded (already scaled)
    # masks of a reference run instead of drawing from the device generator
    REPLAY = None

    def __init__(self, drop_prob: float = 0.):
        super().__init__()
        self.drop_prob = drop_prob

    def keep_mask(self, x):
        if self.drop_prob == 0. or not self.training:
            return None
        keep_prob = 1 - self.drop_prob
        shape = (x.shape[0],) + (1,) * (x.ndim - 1)
        if DropPath.REPLAY is not None:
            return DropPath.REPLAY.pop(0).to(device=x.device, dtype=torch.float32).reshape(shape)
        # fp32 like upstream, where the masked tensor is the fp32 LayerScale output under autocast
        m = torch.empty(shape, dtype=torch.float32, device=x.device).bernoulli_(keep_prob)
        if keep_prob > 0.0:
            m.div_(keep_prob)
        return m

    def forward(self, x):
        m = self.keep_mask(x)
        return x if m is None else x * m


class Attention(nn.Module):
    """Multi-head self-attention (reference vision_transformer.py:145-197); unmasked SDPA over <= 769 tokens."""

    def __init__(self, dim: int, num_heads: int = 8, qkv_bias: bool = False, **kwargs):
        super().__init__()
        assert dim % num_heads == 0, 'dim should be divisible by num_heads'
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.scale = self.head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.q_norm = nn.Identity()
        self.k_norm = nn.Identity()
        self.attn_drop = nn.Dropout(0.)
        self.proj = nn.Linear(dim, dim)
        self.proj_drop = nn.Dropout(0.)

    def forward(self, x, attn_mask=None):
        if attn_mask is not None:
            raise NotImplementedError("attention masks (lat_lora tuning) are outside the hot path")
        qkv = nn_ops.linear(x, self.qkv.weight, self.qkv.bias)
        x = nn_ops.attention_qkvpacked(qkv, self.num_heads)
        return nn_ops.linear(x, self.proj.weight, self.proj.bias)


class LayerScale(nn.Module):
    def __init__(self, dim: int, init_values: float = 1e-5, inplace: bool = False):
        super().__init__()
        self.inplace = inplace
        self.gamma = nn.Parameter(init_values * torch.ones(dim))

    def forward(self, x):
        return x * self.gamma


class Block(nn.Module):
    """Pre-LN transformer block with LayerScale and DropPath (reference :295-339)."""

    def __init__(self, dim, num_heads, mlp_ratio=4., qkv_bias=False, init_values=None, drop_path=0., eps=1e-6):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim, eps=eps)
        self.attn = Attention(dim, num_heads=num_heads, qkv_bias=qkv_bias)
        self.ls1 = LayerScale(dim, init_values=init_values) if init_values else nn.Identity()
        self.drop_path1 = DropPath(drop_path) if drop_path > 0. else nn.Identity()
        self.norm2 = nn.LayerNorm(dim, eps=eps)
        self.mlp = Mlp(in_features=dim, hidden_features=int(dim * mlp_ratio))
        self.ls2 = LayerScale(dim, init_values=init_values) if init_values else nn.Identity()
        self.drop_path2 = DropPath(drop_path) if drop_path > 0. else nn.Identity()

    def _branch(self, x, y, ls, dp):
        # x + drop_path(layer_scale(y)) fused into one elementwise op
        gamma = ls.gamma if isinstance(ls, LayerScale) else None
        mask = dp.keep_mask(y) if isinstance(dp, DropPath) else None
        return nn_ops.residual_scale_add(x, y, gamma, mask)

    def forward(self, x, attn_mask=None):
        y = self.attn(nn_ops.layer_norm(x, self.norm1.weight, self.norm1.bias, self.norm1.eps), attn_mask)
        x = self._branch(x, y, self.ls1, self.drop_path1)
        y = self.mlp(nn_ops.layer_norm(x, self.norm2.weight, self.norm2.bias, self.norm2.eps))
        return self._branch(x, y, self.ls2, self.drop_path2)


class VisionTransformer(nn.Module):
    """The subset of timm's VisionTransformer the tokenizer uses: class token, learned abs pos-embed (with the
    dynamic resample used for the latent grid), pre-LN blocks, final LayerNorm, no classifier head
    (num_classes = 0 in the dinov2 pretrained cfg)."""

    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4.,
                 qkv_bias=True, init_values=None, drop_path_rate=0., num_latent_tokens=32, attn_layer=None, **unused):
        super().__init__()
        self.num_features = self.embed_dim = embed_dim
        self.num_prefix_tokens = 1
        self.num_reg_tokens = 0
        self.has_class_token = True
        self.no_embed_class = False
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim)
        num_patches = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.reg_token = None
        self.pos_embed = nn.Parameter(torch.randn(1, num_patches + 1, embed_dim) * .02)
        self.pos_drop = nn.Dropout(0.)
        self.patch_drop = nn.Identity()
        self.norm_pre = nn.Identity()
        dpr = [x.item() for x in torch.linspace(0, drop_path_rate, depth)]
        self.blocks = nn.Sequential(*[
            Block(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, init_values=init_values,
                  drop_path=dpr[i]) for i in range(depth)])
        self.norm = nn.LayerNorm(embed_dim, eps=1e-6)
        self.fc_norm = nn.Identity()
        self.head_drop = nn.Dropout(0.)
        self.head = nn.Identity()
        self.init_weights()

    def init_weights(self):
        trunc_normal_(self.pos_embed, std=.02)
        nn.init.normal_(self.cls_token, std=1e-6)
        for m in self.modules():  # init_weights_vit_timm
            if isinstance(m, nn.Linear):
                trunc_normal_(m.weight, std=.02)
                if m.bias is not None:
                    nn.init.zeros_(m.bias)

    def no_weight_decay(self):
        return {'pos_embed', 'cls_token', 'dist_token'}

    def _pos_embed(self, x):
        """reference :818-851 — 3-D input (B,L,C): add the table as is; 4-D input (B,H,W,C): resample the grid part."""
        if x.dim() == 4:
            B, H, W, C = x.shape
            pos_embed = resample_abs_pos_embed(self.pos_embed, (H, W), num_prefix_tokens=self.num_prefix_tokens)
            x = x.reshape(B, -1, C)
        else:
            pos_embed = self.pos_embed
        x = torch.cat([self.cls_token.expand(x.shape[0], -1, -1), x], dim=1)
        return x + pos_embed

    def forward_features(self, x, input_affine=None):
        x = self.patch_embed(x) if input_affine is None else self.patch_embed(x, affine=input_affine)
        from .. import ops_dense
        if x.dim() == 3 and ops_dense.token_assemble_supported(x, x.shape[-1]):
            # [cls | patches] + position table in one kernel (ops_dense.TokenAssembleFn); the sample-independent part comes from _pos_embed
            # on ONE zero sample — for the frozen semantic teacher it is a constant, cached per parameter version
            key = (self.cls_token._version, self.pos_embed._version, self.pos_embed.data_ptr(), x.shape[1])
            frozen = not (self.cls_token.requires_grad or self.pos_embed.requires_grad)
            cached = getattr(self, "_xq_token_table", None)
            if frozen and cached is not None and cached[0] == key:
                table = cached[1]
            else:
                table = self._pos_embed(torch.zeros(1, x.shape[1], x.shape[2], dtype=torch.float32, device=x.device))
                if frozen:
                    self._xq_token_table = (key, table.detach())
            # (no cast to the autocast dtype here: upstream's plain VisionTransformer hands the fp32 sum to the blocks, whose LayerNorm
            # autocast runs in fp32 — the kernel's rounding switch stays off)
            x = ops_dense.TokenAssembleFn.apply(x, table, self.num_prefix_tokens, False)
        else:
            x = self._pos_embed(x)
        return nn_ops.vit_blocks(self.blocks, x, self.norm)

    def forward(self, x, input_affine=None):
        x = self.forward_features(x, input_affine=input_affine)
        return x[:, 0]  # global_pool == 'token', head == Identity


_DINOV2_GEOMETRY = {  # reference vision_transformer.py:2886-2931 (patch_size/img_size come from model_kwargs)
    'vit_small_patch14_dinov2': dict(embed_dim=384, depth=12, num_heads=6, init_values=1e-5),
    'vit_base_patch14_dinov2': dict(embed_dim=768, depth=12, num_heads=12, init_values=1e-5),
    'vit_large_patch14_dinov2': dict(embed_dim=1024, depth=24, num_heads=16, init_values=1e-5),
}


def create_model(model_name: str, pretrained: bool = False, **kwargs) -> VisionTransformer:
    """Stand-in for timm.create_model for the dinov2 entries the tokenizer accepts (dinov2.py:26-30).
    There is no network / checkpoint cache in this build, so weights are random-init; load reference
    checkpoints with `load_state_dict` (names are timm's)."""
    base = model_name.split('.')[0]
    if base not in _DINOV2_GEOMETRY:
        raise ValueError(f"{model_name} not found (supported: {sorted(_DINOV2_GEOMETRY)})")
    args = dict(patch_size=14, **_DINOV2_GEOMETRY[base])
    args.update(kwargs)
    return VisionTransformer(**args)
