"""Subprocess body of tests/test_timm_shim_pin.py: the reference's vendored VisionTransformer (dino_enc/vision_transformer.py) built
over oracle/timm_shim.py vs HuggingFace transformers' Dinov2Model — an independent implementation of the same architecture
(vit_*_patch14_dinov2) — on identical weights.  transformers must be imported BEFORE the reference loader puts its stubs for the
absent torchvision / timm into sys.modules."""
import sys

import torch

ROOT = sys.argv[1]
sys.path.insert(0, ROOT)
from transformers import Dinov2Config, Dinov2Model  # noqa: E402

D, DEPTH, HEADS, IMG, PATCH = 768, 2, 12, 64, 16
cfg = Dinov2Config(hidden_size=D, num_hidden_layers=DEPTH, num_attention_heads=HEADS, mlp_ratio=4, image_size=IMG, patch_size=PATCH,
                   layerscale_value=1e-5, qkv_bias=True, hidden_act="gelu", layer_norm_eps=1e-6, use_swiglu_ffn=False)
hf = Dinov2Model(cfg).eval()
from oracle.ref_import import load_reference  # noqa: E402
load_reference()
vit = sys.modules["tokenizer.tokenizer_image.dino_enc.vision_transformer"]
torch.manual_seed(0)
m = vit.vit_base_patch14_dinov2(pretrained=False, img_size=IMG, patch_size=PATCH, depth=DEPTH, drop_path_rate=0.0).eval()
sd = m.state_dict()
with torch.no_grad():
    for k in sd:        # O(1) LayerScale / biases so that every layer matters
        sd[k] = 1 + 0.1 * torch.randn_like(sd[k]) if "norm" in k else torch.randn_like(sd[k]) * (0.2 if sd[k].dim() > 1 else 0.5)
    m.load_state_dict(sd)
    h = {"embeddings.cls_token": sd["cls_token"], "embeddings.mask_token": hf.state_dict()["embeddings.mask_token"],
         "embeddings.position_embeddings": sd["pos_embed"], "embeddings.patch_embeddings.projection.weight": sd["patch_embed.proj.weight"],
         "embeddings.patch_embeddings.projection.bias": sd["patch_embed.proj.bias"], "layernorm.weight": sd["norm.weight"],
         "layernorm.bias": sd["norm.bias"]}
    for i in range(DEPTH):
        p, q = f"blocks.{i}.", f"encoder.layer.{i}."
        for n in ("norm1", "norm2"):
            h[q + n + ".weight"], h[q + n + ".bias"] = sd[p + n + ".weight"], sd[p + n + ".bias"]
        w, b = sd[p + "attn.qkv.weight"], sd[p + "attn.qkv.bias"]
        for j, nm in enumerate(("query", "key", "value")):       # timm packs q, k, v rows in this order (vision_transformer.py:173-176)
            h[q + f"attention.attention.{nm}.weight"], h[q + f"attention.attention.{nm}.bias"] = w[j * D:(j + 1) * D], b[j * D:(j + 1) * D]
        h[q + "attention.output.dense.weight"], h[q + "attention.output.dense.bias"] = sd[p + "attn.proj.weight"], sd[p + "attn.proj.bias"]
        h[q + "layer_scale1.lambda1"], h[q + "layer_scale2.lambda1"] = sd[p + "ls1.gamma"], sd[p + "ls2.gamma"]
        for n in ("fc1", "fc2"):
            h[q + f"mlp.{n}.weight"], h[q + f"mlp.{n}.bias"] = sd[p + f"mlp.{n}.weight"], sd[p + f"mlp.{n}.bias"]
    hf.load_state_dict(h, strict=True)
    x = torch.randn(2, 3, IMG, IMG)
    a = m.forward_features(x)
    b = hf(pixel_values=x).last_hidden_state
print("MAXDIFF", (a - b).abs().max().item(), "SCALE", a.abs().max().item())
