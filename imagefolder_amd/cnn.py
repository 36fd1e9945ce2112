"""CNN encoder / decoder of the tokenizer (taming/LlamaGen VQGAN backbone), MI355X path.

Mirrors reference tokenizer/tokenizer_image/xqgan_model.py: Encoder :454-514, Decoder :518-584, ResnetBlock :587-622,
AttnBlock :625-659, Upsample :675-686, Downsample :689-704 — same constructor arguments and parameter names
(`conv_blocks.{l}.res.{i}.conv1.weight`, `...attn.{i}.q.weight`, `...downsample.conv.weight`, `mid.{i}...`), so
reference checkpoints load.  Tensor math goes through imagefolder_amd.nn_ops: GroupNorm(32, eps=1e-6)+SiLU is one
op (the reference runs three), convs and the single-head spatial attention are ops that can be swapped for HIP kernels.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import nn_ops


def _gn(ch):
    return nn.GroupNorm(num_groups=32, num_channels=ch, eps=1e-6, affine=True)


def _norm_act(gn, x, silu=True):
    return nn_ops.group_norm_silu(x, gn.num_groups, gn.weight, gn.bias, gn.eps, silu=silu)


def _conv(c, x):
    return nn_ops.conv2d(x, c.weight, c.bias, stride=c.stride[0], padding=c.padding[0])


class ResnetBlock(nn.Module):
    def __init__(self, in_channels, out_channels=None, conv_shortcut=False, dropout=0.0, norm_type='group'):
        super().__init__()
        assert norm_type == 'group', "SyncBatchNorm variant is unused by the yamls (SURVEY §2.2)"
        out_channels = in_channels if out_channels is None else out_channels
        self.in_channels, self.out_channels = in_channels, out_channels
        self.use_conv_shortcut = conv_shortcut
        self.norm1 = _gn(in_channels)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, 1, 1)
        self.norm2 = _gn(out_channels)
        self.dropout = nn.Dropout(dropout)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, 1, 1)
        if in_channels != out_channels:
            if conv_shortcut:
                self.conv_shortcut = nn.Conv2d(in_channels, out_channels, 3, 1, 1)
            else:
                self.nin_shortcut = nn.Conv2d(in_channels, out_channels, 1, 1, 0)

    def forward(self, x):
        h = _conv(self.conv1, _norm_act(self.norm1, x))
        h = _conv(self.conv2, self.dropout(_norm_act(self.norm2, h)))
        if self.in_channels != self.out_channels:
            x = _conv(self.conv_shortcut if self.use_conv_shortcut else self.nin_shortcut, x)
        return x + h


class AttnBlock(nn.Module):
    """single-head attention over the H*W positions of a feature map (1x1-conv q/k/v/proj)"""

    def __init__(self, in_channels, norm_type='group'):
        super().__init__()
        self.norm = _gn(in_channels)
        self.q = nn.Conv2d(in_channels, in_channels, 1)
        self.k = nn.Conv2d(in_channels, in_channels, 1)
        self.v = nn.Conv2d(in_channels, in_channels, 1)
        self.proj_out = nn.Conv2d(in_channels, in_channels, 1)

    def forward(self, x):
        h = _norm_act(self.norm, x, silu=False)
        b, c, hh, ww = h.shape
        h = nn_ops.spatial_attention(_conv(self.q, h), _conv(self.k, h), _conv(self.v, h))
        return x + _conv(self.proj_out, h)


class Upsample(nn.Module):
    def __init__(self, in_channels, with_conv):
        super().__init__()
        self.with_conv = with_conv
        if with_conv:
            self.conv = nn.Conv2d(in_channels, in_channels, 3, 1, 1)

    def forward(self, x):
        if self.with_conv:
            return nn_ops.conv2d_upsample(x, self.conv.weight, self.conv.bias)
        return F.interpolate(x, scale_factor=2.0, mode="nearest")


class Downsample(nn.Module):
    def __init__(self, in_channels, with_conv):
        super().__init__()
        self.with_conv = with_conv
        if with_conv:
            self.conv = nn.Conv2d(in_channels, in_channels, 3, 2, 0)  # asymmetric (0,1,0,1) pad applied by hand

    def forward(self, x):
        if self.with_conv:
            return nn_ops.conv2d_downsample(x, self.conv.weight, self.conv.bias)
        return F.avg_pool2d(x, kernel_size=2, stride=2)


class _Level(nn.Module):
    pass


def _mid(ch, dropout, norm_type):
    return nn.ModuleList([ResnetBlock(ch, ch, dropout=dropout, norm_type=norm_type), AttnBlock(ch, norm_type=norm_type),
                          ResnetBlock(ch, ch, dropout=dropout, norm_type=norm_type)])


class Encoder(nn.Module):
    def __init__(self, in_channels=3, ch=128, ch_mult=(1, 1, 2, 2, 4), num_res_blocks=2, norm_type='group', dropout=0.0,
                 resamp_with_conv=True, z_channels=256):
        super().__init__()
        self.num_resolutions = len(ch_mult)
        self.num_res_blocks = num_res_blocks
        self.conv_in = nn.Conv2d(in_channels, ch, 3, 1, 1)
        in_ch_mult = (1,) + tuple(ch_mult)
        self.conv_blocks = nn.ModuleList()
        for lvl in range(self.num_resolutions):
            level = _Level()
            level.res, level.attn = nn.ModuleList(), nn.ModuleList()
            block_in, block_out = ch * in_ch_mult[lvl], ch * ch_mult[lvl]
            for _ in range(num_res_blocks):
                level.res.append(ResnetBlock(block_in, block_out, dropout=dropout, norm_type=norm_type))
                block_in = block_out
                if lvl == self.num_resolutions - 1:
                    level.attn.append(AttnBlock(block_in, norm_type))
            if lvl != self.num_resolutions - 1:
                level.downsample = Downsample(block_in, resamp_with_conv)
            self.conv_blocks.append(level)
        self.mid = _mid(block_in, dropout, norm_type)
        self.norm_out = _gn(block_in)
        self.conv_out = nn.Conv2d(block_in, z_channels, 3, 1, 1)

    def forward(self, x):
        h = _conv(self.conv_in, x)
        for lvl, level in enumerate(self.conv_blocks):
            for i in range(self.num_res_blocks):
                h = level.res[i](h)
                if len(level.attn) > 0:
                    h = level.attn[i](h)
            if lvl != self.num_resolutions - 1:
                h = level.downsample(h)
        for blk in self.mid:
            h = blk(h)
        return _conv(self.conv_out, _norm_act(self.norm_out, h))


class Decoder(nn.Module):
    def __init__(self, z_channels=256, ch=128, ch_mult=(1, 1, 2, 2, 4), num_res_blocks=2, norm_type="group", dropout=0.0,
                 resamp_with_conv=True, out_channels=3):
        super().__init__()
        self.num_resolutions = len(ch_mult)
        self.num_res_blocks = num_res_blocks
        block_in = ch * ch_mult[self.num_resolutions - 1]
        self.conv_in = nn.Conv2d(z_channels, block_in, 3, 1, 1)
        self.mid = _mid(block_in, dropout, norm_type)
        self.conv_blocks = nn.ModuleList()
        for lvl in reversed(range(self.num_resolutions)):
            level = _Level()
            level.res, level.attn = nn.ModuleList(), nn.ModuleList()
            block_out = ch * ch_mult[lvl]
            for _ in range(num_res_blocks + 1):
                level.res.append(ResnetBlock(block_in, block_out, dropout=dropout, norm_type=norm_type))
                block_in = block_out
                if lvl == self.num_resolutions - 1:
                    level.attn.append(AttnBlock(block_in, norm_type))
            if lvl != 0:
                level.upsample = Upsample(block_in, resamp_with_conv)
            self.conv_blocks.append(level)
        self.norm_out = _gn(block_in)
        self.conv_out = nn.Conv2d(block_in, out_channels, 3, 1, 1)

    @property
    def last_layer(self):
        return self.conv_out.weight

    def forward(self, z):
        h = _conv(self.conv_in, z)
        for blk in self.mid:
            h = blk(h)
        for lvl, level in enumerate(self.conv_blocks):
            for i in range(self.num_res_blocks + 1):
                h = level.res[i](h)
                if len(level.attn) > 0:
                    h = level.attn[i](h)
            if lvl != self.num_resolutions - 1:
                h = level.upsample(h)
        return _conv(self.conv_out, _norm_act(self.norm_out, h))
