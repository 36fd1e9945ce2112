"""Bulk tokenisation throughput (reference scripts/pretokenization.py:150-259): images/sec of
encoder + quant_conv + fused nearest-code search + pinned double-buffered D2H through imagefolder_amd.tokenize.BulkTokenizer,
at inference batch sizes, bf16 autocast, images resident in HBM (random-init weights: no checkpoints offline).
    python tools/bench_tokenize.py [--config VQ-8192] [--batch 256] [--batches 12] [--out gpurun_out/bulk_tokenize.jsonl]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (the BASELINE config table)
from imagefolder_amd import tokenize as tk  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", nargs="*", default=["VQ-8192", "VP2-16384", "MSVR10P2-4096"])
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--batches", type=int, default=12)
    ap.add_argument("--augment", default="flip", choices=["flip", "none"])
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    dev = torch.device("cuda")
    from imagefolder_amd.xqgan_model import VQ_models
    lines = []
    for name in a.config:
        c = bench.CONFIGS[name]
        torch.manual_seed(0)
        model = VQ_models["VQ-16"](codebook_size=c["V"], codebook_embed_dim=c["C"], v_patch_nums=list(c["pns"]), enc_type=c["enc"], dec_type=c["enc"],
                                   semantic_guide="none", detail_guide="none", num_latent_tokens=c["L"],
                                   encoder_model="vit_base_patch14_dinov2.lvd142m", decoder_model="vit_base_patch14_dinov2.lvd142m",
                                   abs_pos_embed=True, product_quant=c["P"], share_quant_resi=4, half_sem=c["half_sem"]).to(dev).eval()
        g = torch.Generator(device=dev).manual_seed(7)
        B = a.batch // (2 if a.augment == "flip" else 1)          # the flip doubles the batch the tokenizer sees (pretokenization.py:227-228)
        imgs = [torch.rand(B, 3, 256, 256, device=dev, generator=g) * 2 - 1 for _ in range(2)]
        labels = torch.arange(B) % 1000

        def batches(n):
            for i in range(n):
                yield imgs[i & 1], labels
        tk.BulkTokenizer(model, amp_dtype=torch.bfloat16, augment=a.augment).run(batches(2))          # warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        bt = tk.BulkTokenizer(model, amp_dtype=torch.bfloat16, augment=a.augment).run(batches(a.batches))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        cls, tok = bt.records
        n_views = len(cls)
        line = {"metric": "bulk tokenisation, tokenizer forward passes (images incl. flipped views) per second", "value": n_views / dt,
                "unit": "images/sec", "config": name, "tokens_per_image": int(tok.shape[1]), "views": n_views,
                "batch_seen_by_the_tokenizer": a.batch, "augment": a.augment, "dtype": "bf16 autocast",
                "what": "encoder + quant_conv + fused nearest-code kernels + pinned double-buffered D2H of the int64 tokens; images resident in HBM"}
        print(json.dumps(line), flush=True)
        lines.append(line)
        del model, bt
        torch.cuda.empty_cache()
    if a.out:
        os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
        with open(a.out, "w") as f:
            for l in lines:
                f.write(json.dumps(l) + "\n")


if __name__ == "__main__":
    main()
