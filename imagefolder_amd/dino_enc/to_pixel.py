"""Decoder tail: tokens -> pixels.  Mirrors reference dino_enc/to_pixel.py:36-95 ('linear' head only — the yamls
never select 'conv'/'siren'/'identity')."""
import torch
import torch.nn as nn

from .. import nn_ops


class ToPixel(nn.Module):
    def __init__(self, to_pixel='linear', img_size=256, in_channels=3, in_dim=512, patch_size=16) -> None:
        super().__init__()
        if to_pixel != 'linear':
            raise NotImplementedError("only the 'linear' pixel head is on the hot path (DINOv2Decoder default)")
        self.to_pixel_name = to_pixel
        self.patch_size = patch_size
        self.num_patches = (img_size // patch_size) ** 2
        self.in_channels = in_channels
        self.model = nn.Linear(in_dim, in_channels * patch_size * patch_size)

    def get_last_layer(self):
        return self.model.weight

    def unpatchify(self, x):
        """x: (N, L, p*p*3) -> imgs (N, 3, H, W); token = (h, w), feature = (p, q, c)  (to_pixel.py:70-81)"""
        p = self.patch_size
        h = w = int(x.shape[1] ** .5)
        assert h * w == x.shape[1]
        x = x.reshape(x.shape[0], h, w, p, p, 3)
        x = torch.einsum('nhwpqc->nchpwq', x)
        return x.reshape(x.shape[0], 3, h * p, h * p)

    def forward(self, x):
        return self.unpatchify(nn_ops.linear(x, self.model.weight, self.model.bias))
