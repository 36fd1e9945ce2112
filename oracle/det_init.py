"""TEST INFRASTRUCTURE — deterministic, structure-independent weights for model-level parity fixtures.

Checkpoints are not reachable offline and full-size weights (72-258 M parameters) cannot be committed, so the
model-level goldens store only inputs/outputs; both sides (the reference model in oracle/make_golden.py, the mirror in
the tests) fill their state_dict from this function: every tensor is drawn from its own generator seeded with
crc32(name), so the values depend on the parameter NAME and SHAPE only, not on construction order or module types."""
import zlib

import torch


def det_state_dict(sd, seed=0):
    out = {}
    for name, t in sd.items():
        if not t.dtype.is_floating_point:
            out[name] = t.clone()
            continue
        g = torch.Generator().manual_seed((zlib.crc32(name.encode()) + 7919 * seed) & 0x7FFFFFFF)
        r = torch.randn(t.shape, generator=g, dtype=torch.float32)
        leaf = name.split(".")[-1]
        if "norm" in name and leaf == "weight" and t.dim() == 1:
            v = 1.0 + 0.05 * r                       # norm scales around 1
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
