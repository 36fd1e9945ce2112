"""TEST INFRASTRUCTURE — deterministic, structure-independent weights for model-level parity fixtures.

Checkpoints are not reachable offline and full-size weights (72-258 M parameters) cannot be committed, so the
model-level goldens store only inputs/outputs; both sides (the reference model in oracle/make_golden.py, the mirror in
the tests) fill their state_dict from this function: every tensor is drawn from its own generator seeded with
crc32(name), so the values depend on the parameter NAME and SHAPE only, not on construction order or module types."""
import re
import zlib

import torch

# input-normalisation constants registered as buffers (LPIPS ScalingLayer, FrozenDINOSmallNoDrop): keep the constructor's values
_CONSTANT_BUFFERS = ("scaling_layer.shift", "scaling_layer.scale", "perceptual_loss.shift", "perceptual_loss.scale", "x_scale", "x_shift")
_HEAD_NORM_SCALE = re.compile(r"heads\.\d+\.(0|1\.fn)\.1\.weight$")


def det_state_dict(sd, seed=0):
    out = {}
    for name, t in sd.items():
        if not t.dtype.is_floating_point or name.endswith(_CONSTANT_BUFFERS):
            out[name] = t.clone()
            continue
        g = torch.Generator().manual_seed((zlib.crc32(name.encode()) + 7919 * seed) & 0x7FFFFFFF)
        r = torch.randn(t.shape, generator=g, dtype=torch.float32)
        leaf = name.split(".")[-1]
        if "norm" in name and leaf == "weight" and t.dim() == 1:
            v = 1.0 + 0.05 * r                       # norm scales around 1
        elif _HEAD_NORM_SCALE.search(name):
            v = 1.0 + 0.1 * r                        # BatchNormLocal scales of the DinoDisc heads (Sequential index 1 of a make_block)
        elif leaf == "gamma":
            v = 0.2 + 0.02 * r                       # LayerScale: large enough for the blocks to matter (init is 1e-5)
        elif any(k in name for k in ("latent_tokens", "pos_embed", "cls_token", "mask_token", "lvl_embed")):
            v = 0.5 * r                              # token / position tables: O(1) entries so tokens differ
        elif "attn.qkv.weight" in name:
            v = r * (2.0 / t.shape[1] ** 0.5)        # sharper attention than a variance-preserving init
        elif leaf == "bias" or t.dim() <= 1:
            v = 0.02 * r
        elif "embedding" in name:
            v = torch.nn.functional.normalize(r, dim=-1)   # unit-norm codebook rows like the reference init
        elif "ema_vocab_hit" in name:
            v = torch.zeros_like(r)
        else:
            fan_in = t[0].numel() if t.dim() > 1 else t.numel()
            v = r * (1.0 / max(1.0, fan_in) ** 0.5)   # variance-preserving for convs / linears / token tables
        out[name] = v.to(t.dtype)
    return out


def vqloss_inputs(B, seed, size=224):
    """images, the pre-image of the reconstruction and the 1x1 'last layer' that maps it to the reconstruction (so that the adaptive weight
    has a last_layer to differentiate to, vq_loss.py:153-159); 224 x 224: DinoDisc's trunk neither crops nor resizes (discriminator_dino.py:330-336)"""
    g = torch.Generator().manual_seed(9000 + seed)
    imgs = torch.rand(B, 3, size, size, generator=g) * 2 - 1
    pre = (imgs + 0.3 * torch.randn(B, 3, size, size, generator=g)).clamp(-1.2, 1.2)
    last = torch.eye(3).view(3, 3, 1, 1) + 0.1 * torch.randn(3, 3, 1, 1, generator=g)
    return imgs, pre, last
