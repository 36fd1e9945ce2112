"""DINOv2-ViT encoder / decoder wrappers of the tokenizer.

Mirror of reference tokenizer/tokenizer_image/dino_enc/dinov2.py (DINOv2Encoder :18-198, DINOv2Decoder :201-365):
same constructor arguments, parameter/buffer names (`model.*`, `latent_tokens`, `lvl_embed`, `lvl1LC`, `mask_token`,
`latent_pos_embed`, `to_pixel.model.*`) and token layouts.  Only `tuning_method` 'full' and
'frozen' exist here: LoRA variants need `peft`, which is outside the hot path (all five yamls use 'full').
"""
import math

import torch
import torch.nn as nn

from .. import nn_ops
from .to_pixel import ToPixel
from .vision_transformer import create_model, trunc_normal_

_DINOV2_NAMES = ['vit_small_patch14_dinov2.lvd142m', 'vit_base_patch14_dinov2.lvd142m',
                 'vit_large_patch14_dinov2.lvd142m']


def _apply_tuning(module, tuning_method):
    if tuning_method == 'full':
        return
    if tuning_method == 'frozen':
        for p in module.model.parameters():
            p.requires_grad = False
        return
    raise NotImplementedError(f"tuning_method={tuning_method!r} needs peft (LoRA); only 'full'/'frozen' are mirrored")


class DINOv2Encoder(nn.Module):
    def __init__(self, in_channels=3, num_latent_tokens=32, use_attn_mask=False,
                 model_name='vit_small_patch14_dinov2.lvd142m',
                 model_kwargs={'img_size': 224, 'patch_size': 14, 'drop_path_rate': 0.0, },
                 pretrained=True, tuning_method='lora', tuning_kwargs={'r': 8}, abs_pos_embed=False, product_quant=1):
        super().__init__()
        assert model_name in _DINOV2_NAMES, f"{model_name} not found"
        if use_attn_mask:
            raise NotImplementedError("use_attn_mask belongs to the lat_lora tuning path (not mirrored)")
        self.num_latent_tokens = num_latent_tokens
        self.use_attn_mask = False
        self.product_quant = product_quant
        model = create_model(model_name, pretrained=pretrained, **model_kwargs)
        self.embed_dim = model.embed_dim
        self.num_img_tokens = model.patch_embed.num_patches
        self.num_prefix_tokens = model.num_prefix_tokens
        self.abs_pos_embed = abs_pos_embed
        self.model = model
        _apply_tuning(self, tuning_method)

        if self.num_latent_tokens:
            self.latent_tokens = nn.Parameter(torch.zeros(1, self.num_latent_tokens, model.embed_dim))
            nn.init.normal_(self.latent_tokens, std=1e-6)
            if self.abs_pos_embed:
                patch_size = model_kwargs['patch_size']
                n_lvl = 1 + self.product_quant if self.product_quant > 1 else 2
                self.lvl_embed = nn.Embedding(n_lvl, model.embed_dim)
                nn.init.trunc_normal_(self.lvl_embed.weight.data, mean=0, std=math.sqrt(1 / model.embed_dim / 3))
                per = self.num_latent_tokens // self.product_quant
                # NB upstream sizes the image part as patch_size**2 + 1 (= 257 only because 256/16 = 16 = patch_size)
                lvl1LC = torch.cat([torch.full((patch_size * patch_size + 1,), 0)] +
                                   [torch.full((per,), i + 1) for i in range(self.product_quant)]).view(1, -1)
                self.register_buffer('lvl1LC', lvl1LC)
            else:
                self.latent_pos_embed = nn.Parameter(torch.zeros(1, self.num_latent_tokens, model.embed_dim))
                trunc_normal_(self.latent_pos_embed, std=.02)

    def finetine(self, tuning_method, tuning_kwargs={'r': 8}):  # (sic) upstream spelling
        _apply_tuning(self, tuning_method)

    def no_weight_decay(self):
        return ['model.pos_embed', 'model.cls_token', 'model.dist_token', 'latent_tokens', 'latent_pos_embed']

    def _assemble(self, x):
        """patch tokens (B, n, D) fp32 -> the sequence entering the blocks: [cls | patches | latent tokens] + position / level embeddings
        (upstream :149-175, fp32, autocast off)"""
        m = self.model
        with torch.autocast(device_type=x.device.type, enabled=False):
            x = m._pos_embed(x.float())
            if self.num_latent_tokens:
                z = self.latent_tokens.expand(x.size(0), -1, -1)
                if self.abs_pos_embed:
                    P = self.product_quant
                    H = W = int(math.sqrt(self.num_latent_tokens // P))
                    assert H * W == self.num_latent_tokens // P
                    z_list = z.view(x.size(0), P * H, W, -1).chunk(chunks=P, dim=1)
                    z_list = [m._pos_embed(zz)[:, 1:, ] for zz in z_list]  # cls stripped (:164,171)
                    x = torch.cat([x, ] + z_list, dim=1)
                    x = x + self.lvl_embed(self.lvl1LC)  # (1, L, D) broadcast over the batch (upstream expands the index first)
                else:
                    x = torch.cat([x, z + self.latent_pos_embed], dim=1)
        return x

    def forward(self, x, masks=None):
        m = self.model
        x = m.patch_embed(x)
        from .. import ops_dense
        if ops_dense.token_assemble_supported(x, x.shape[-1]):
            # everything of the sequence that does not depend on the sample — class token, position tables, the learnable latent tokens, level
            # embedding — from the op chain above on a batch of ONE (autograd intact: their gradients flow through it), the patch tokens added
            # in one kernel that also applies upstream's cast to the autocast dtype (:177-179), as a rounding kept in fp32
            table = self._assemble(torch.zeros(1, x.shape[1], x.shape[2], dtype=torch.float32, device=x.device))
            x = ops_dense.TokenAssembleFn.apply(x, table, m.num_prefix_tokens, True)
        else:
            # position-embedding section runs in fp32 like upstream (dinov2.py:151 disables autocast here)
            x = self._assemble(x)
            if torch.is_autocast_enabled() and x.is_cuda:
                x = x.to(torch.get_autocast_dtype('cuda'))  # upstream probes the matmul dtype (:177-179)
        x = nn_ops.vit_blocks(m.blocks, x, m.norm)
        if self.num_latent_tokens:
            return x[:, -self.num_latent_tokens:]
        return x[:, self.num_prefix_tokens:]


class DINOv2Decoder(nn.Module):
    def __init__(self, in_channels=3, model_name='vit_small_patch14_dinov2.lvd142m',
                 model_kwargs={'img_size': 224, 'patch_size': 14, 'drop_path_rate': 0.0}, pretrained=True,
                 tuning_method='lora', tuning_kwargs={'r': 8}, num_latent_tokens=32, to_pixel='linear', use_rope=False,
                 cond_latent=False, abs_pos_embed=False):
        super().__init__()
        assert model_name in _DINOV2_NAMES, f"{model_name} not found"
        if use_rope or cond_latent:
            raise NotImplementedError("use_rope / cond_latent are never enabled by VQModel (xqgan_model.py:115-118)")
        model_kwargs = dict(model_kwargs)
        model_kwargs['num_latent_tokens'] = num_latent_tokens
        model = create_model(model_name, pretrained=pretrained, **model_kwargs)
        self.use_rope = False
        self.embed_dim = model.embed_dim
        self.num_img_tokens = model.patch_embed.num_patches
        self.num_prefix_tokens = model.num_prefix_tokens
        self.num_latent_tokens = num_latent_tokens
        self.abs_pos_embed = abs_pos_embed
        self.model = model
        _apply_tuning(self, tuning_method)

        self.mask_token = nn.Parameter(torch.zeros(1, 1, model.embed_dim))
        nn.init.normal_(self.mask_token, std=1e-6)
        if self.abs_pos_embed:
            self.lvl_embed = nn.Embedding(2, model.embed_dim)
            patch_size = model_kwargs['patch_size']
            nn.init.trunc_normal_(self.lvl_embed.weight.data, mean=0, std=math.sqrt(1 / model.embed_dim / 3))
            lvl1LC = torch.cat([torch.full((patch_size * patch_size + 1,), 0),
                                torch.full((self.num_latent_tokens + 1,), 1)]).view(1, -1)
            self.register_buffer('lvl1LC', lvl1LC)
        else:
            self.latent_pos_embed = nn.Parameter(torch.zeros(1, self.num_latent_tokens, model.embed_dim))
            trunc_normal_(self.latent_pos_embed, std=.02)
        self.to_pixel = ToPixel(to_pixel=to_pixel, img_size=model_kwargs['img_size'], in_channels=in_channels,
                                in_dim=model.embed_dim, patch_size=model_kwargs['patch_size'])
        self.cond_latent = False
        # the decoder never embeds pixels: upstream deletes these two parameters (:287-288)
        del self.model.patch_embed.proj.bias
        del self.model.patch_embed.proj.weight

    def finetine(self, tuning_method, tuning_kwargs={'r': 8}):
        _apply_tuning(self, tuning_method)

    def no_weight_decay(self):
        return ['model.pos_embed', 'model.cls_token', 'model.dist_token', 'mask_token', 'latent_pos_embed']

    @property
    def last_layer(self):
        return self.to_pixel.model.weight

    def _assemble(self, z):
        """latent tokens (B, L, D) -> [cls | mask tokens | (cls') | latent tokens] + position / level embeddings (upstream :313-344, fp32)"""
        m = self.model
        x = self.mask_token.expand(z.size(0), self.num_img_tokens, -1)
        with torch.autocast(device_type=z.device.type, enabled=False):
            x = m._pos_embed(x.float())
            if self.abs_pos_embed:
                H = W = int(math.sqrt(self.num_latent_tokens))
                assert H * W == self.num_latent_tokens
                z = m._pos_embed(z.float().view(x.size(0), H, W, -1))  # NOT cls-stripped here (:329-330) -> L+1 tokens
            else:
                z = z.float() + self.latent_pos_embed
            x = torch.cat([x, z], dim=1)
            if self.abs_pos_embed:
                x = x + self.lvl_embed(self.lvl1LC)  # (1, L, D) broadcast over the batch (upstream expands the index first)
        return x

    def forward(self, z):
        m = self.model
        from .. import ops_dense
        if z.dim() == 3 and ops_dense.token_assemble_supported(z, z.shape[-1]):
            # the sample-independent part (class tokens, mask tokens, position tables, level embedding) from the op chain on a batch of ONE
            # zero latent; the quantised latents land behind it in one kernel (TokenAssembleFn, see DINOv2Encoder.forward)
            table = self._assemble(torch.zeros(1, z.shape[1], z.shape[2], dtype=torch.float32, device=z.device))
            start = self.num_img_tokens + m.num_prefix_tokens + (m.num_prefix_tokens if self.abs_pos_embed else 0)
            x = ops_dense.TokenAssembleFn.apply(z, table, start, True)
        else:
            x = self._assemble(z)
            if torch.is_autocast_enabled() and x.is_cuda:
                x = x.to(torch.get_autocast_dtype('cuda'))
        x = nn_ops.vit_blocks(m.blocks, x, m.norm)
        x = x[:, self.num_prefix_tokens:self.num_img_tokens + self.num_prefix_tokens]
        return self.to_pixel(x)
