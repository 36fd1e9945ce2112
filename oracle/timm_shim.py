"""TEST INFRASTRUCTURE — minimal stand-in for the `timm` (==1.0.9, environment.yml:102) symbols the reference's
vendored dino_enc/vision_transformer.py and dino_enc/dinov2.py import (vision_transformer.py:43-51, dinov2.py:7-8).
timm is a pip dependency that is absent from /root/reference and from this image, so its *published* layer
behaviour is restated here; with these in sys.modules the reference's own VisionTransformer / DINOv2Encoder /
DINOv2Decoder classes import and run UNMODIFIED on CPU (random init: the pretrained DINOv2 weights are not
reachable offline).  Used only by oracle/ref_import.py -> oracle/make_golden.py.
"""
import math
import sys
import types
from functools import partial
from unittest.mock import MagicMock

import torch
import torch.nn as nn
import torch.nn.functional as F


# ---- timm.layers ------------------------------------------------------------------------------------------
def trunc_normal_(tensor, mean=0., std=1., a=-2., b=2.):
    return nn.init.trunc_normal_(tensor, mean=mean, std=std, a=a, b=b)


class PatchEmbed(nn.Module):
    def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, norm_layer=None, flatten=True,
                 output_fmt=None, bias=True, strict_img_size=True, dynamic_img_pad=False):
        super().__init__()
        self.patch_size = (patch_size, patch_size) if isinstance(patch_size, int) else tuple(patch_size)
        self.img_size = (img_size, img_size) if isinstance(img_size, int) else tuple(img_size)
        self.grid_size = tuple(s // p for s, p in zip(self.img_size, self.patch_size))
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.flatten = flatten
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=self.patch_size, stride=self.patch_size, bias=bias)
        self.norm = norm_layer(embed_dim) if norm_layer else nn.Identity()

    def forward(self, x):
        x = self.proj(x)
        if self.flatten:
            x = x.flatten(2).transpose(1, 2)  # NCHW -> NLC
        return self.norm(x)


class Mlp(nn.Module):
    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, norm_layer=None,
                 bias=True, drop=0., use_conv=False):
        super().__init__()
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        self.fc1 = nn.Linear(in_features, hidden_features, bias=bias)
        self.act = act_layer()
        self.drop1 = nn.Dropout(drop)
        self.norm = norm_layer(hidden_features) if norm_layer is not None else nn.Identity()
        self.fc2 = nn.Linear(hidden_features, out_features, bias=bias)
        self.drop2 = nn.Dropout(drop)

    def forward(self, x):
        return self.drop2(self.fc2(self.norm(self.drop1(self.act(self.fc1(x))))))


class DropPath(nn.Module):
    # golden generation: when RECORD is a list, every mask drawn (after the 1/keep scaling) is appended to it, in call order
    RECORD = None

    def __init__(self, drop_prob: float = 0., scale_by_keep: bool = True):
        super().__init__()
        self.drop_prob = drop_prob
        self.scale_by_keep = scale_by_keep

    def forward(self, x):
        if self.drop_prob == 0. or not self.training:
            return x
        keep_prob = 1 - self.drop_prob
        shape = (x.shape[0],) + (1,) * (x.ndim - 1)
        random_tensor = x.new_empty(shape).bernoulli_(keep_prob)
        if keep_prob > 0.0 and self.scale_by_keep:
            random_tensor.div_(keep_prob)
        if DropPath.RECORD is not None:
            DropPath.RECORD.append(random_tensor.detach().reshape(-1).clone())
        return x * random_tensor


def resample_abs_pos_embed(posemb, new_size, old_size=None, num_prefix_tokens: int = 1, interpolation: str = 'bicubic',
                           antialias: bool = True, verbose: bool = False):
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
        posemb_prefix, posemb = None, posemb
    embed_dim = posemb.shape[-1]
    orig_dtype = posemb.dtype
    posemb = posemb.float()
    posemb = posemb.reshape(1, old_size[0], old_size[1], -1).permute(0, 3, 1, 2)
    posemb = F.interpolate(posemb, size=new_size, mode=interpolation, antialias=antialias)
    posemb = posemb.permute(0, 2, 3, 1).reshape(1, -1, embed_dim)
    posemb = posemb.to(orig_dtype)
    if posemb_prefix is not None:
        posemb = torch.cat([posemb_prefix, posemb], dim=1)
    return posemb


def use_fused_attn(experimental: bool = False) -> bool:
    return hasattr(F, 'scaled_dot_product_attention')


def get_norm_layer(norm_layer):
    return norm_layer  # None -> caller's default (partial(nn.LayerNorm, eps=1e-6))


def get_act_layer(act_layer):
    return act_layer


# ---- timm.models._manipulate / _builder / _registry ---------------------------------------------------------
def named_apply(fn, module: nn.Module, name='', depth_first: bool = True, include_root: bool = False):
    if not depth_first and include_root:
        fn(module=module, name=name)
    for child_name, child_module in module.named_children():
        child_name = '.'.join((name, child_name)) if name else child_name
        named_apply(fn=fn, module=child_module, name=child_name, depth_first=depth_first, include_root=True)
    if depth_first and include_root:
        fn(module=module, name=name)
    return module


def build_model_with_cfg(model_cls, variant, pretrained, pretrained_filter_fn=None, pretrained_strict=True,
                         feature_cfg=None, **kwargs):
    assert not pretrained, "no pretrained weights offline"
    if 'dinov2' in variant:
        kwargs.setdefault('num_classes', 0)  # pretrained_cfg of the *.lvd142m entries
    return model_cls(**kwargs)


def register_model(fn):
    return fn


def install(ref_vit_module_name='tokenizer.tokenizer_image.dino_enc.vision_transformer'):
    """Puts the shim into sys.modules (replacing MagicMock placeholders)."""
    timm = types.ModuleType('timm')
    layers = types.ModuleType('timm.layers')
    for n, v in dict(PatchEmbed=PatchEmbed, Mlp=Mlp, DropPath=DropPath, trunc_normal_=trunc_normal_,
                     resample_abs_pos_embed=resample_abs_pos_embed, use_fused_attn=use_fused_attn,
                     get_act_layer=get_act_layer, get_norm_layer=get_norm_layer).items():
        setattr(layers, n, v)
    for n in ['AttentionPoolLatent', 'RmsNorm', 'PatchDropout', 'SwiGLUPacked', 'lecun_normal_', 'resample_patch_embed',
              'LayerType']:
        setattr(layers, n, MagicMock())
    data = types.ModuleType('timm.data')
    data.IMAGENET_DEFAULT_MEAN = (0.485, 0.456, 0.406)
    data.IMAGENET_DEFAULT_STD = (0.229, 0.224, 0.225)
    data.IMAGENET_INCEPTION_MEAN = (0.5, 0.5, 0.5)
    data.IMAGENET_INCEPTION_STD = (0.5, 0.5, 0.5)
    data.OPENAI_CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
    data.OPENAI_CLIP_STD = (0.26862954, 0.26130258, 0.27577711)
    models = types.ModuleType('timm.models')
    builder = types.ModuleType('timm.models._builder')
    builder.build_model_with_cfg = build_model_with_cfg
    features = types.ModuleType('timm.models._features')
    features.feature_take_indices = MagicMock()
    manipulate = types.ModuleType('timm.models._manipulate')
    manipulate.named_apply = named_apply
    manipulate.checkpoint_seq = MagicMock()
    manipulate.adapt_input_conv = MagicMock()
    registry = types.ModuleType('timm.models._registry')
    registry.generate_default_cfgs = lambda cfgs: cfgs
    registry.register_model = register_model
    registry.register_model_deprecations = lambda *a, **k: None

    def create_model(model_name, pretrained=False, **kwargs):
        vit = sys.modules[ref_vit_module_name]
        fn = getattr(vit, model_name.split('.')[0])
        return fn(pretrained=False, **kwargs)  # weights are not reachable offline: random init

    models.create_model = create_model
    models.safe_model_name = lambda n: n
    timm.layers, timm.data, timm.models = layers, data, models
    sys.modules.update({'timm': timm, 'timm.layers': layers, 'timm.data': data, 'timm.models': models,
                        'timm.models._builder': builder, 'timm.models._features': features,
                        'timm.models._manipulate': manipulate, 'timm.models._registry': registry})
