"""TEST INFRASTRUCTURE — generates tests/golden/*.npz by running the UNMODIFIED reference modules
(imported from /root/reference through oracle/ref_import.py) on CPU in fp32.

The reference ships no tests / golden vectors for this path, so these fixtures are
the pin: inputs + the reference's own outputs (indices, z_q, losses, usage, autograd grads).
Run here (build container) only:   python oracle/make_golden.py
The GPU box never needs /root/reference: tests read the committed .npz files.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle.ref_import import load_reference  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def meta():
    return dict(torch_version=torch.__version__, generator="oracle/make_golden.py",
                reference="lxa9867/ImageFolder @ /root/reference (2025-04-18 snapshot)")


def gen_vq(name, V, C, B, H, W, seed, codebook_norm=True, beta=0.25, z_scale=1.0, clustered=False):
    """VectorQuantizer.forward/backward + f_to_idxBl_or_fhat (xqgan_model.py:722-833)."""
    R = load_reference()
    torch.manual_seed(seed)
    q = R["VectorQuantizer"](V, C, beta, codebook_norm).train()
    z = torch.randn(B, C, H, W) * z_scale
    if clustered:
        # realistic failure mode: many tokens compete for few codes (low usage at init)
        centers = q.embedding.weight.detach()[torch.randint(0, V, (8,))]
        pick = torch.randint(0, 8, (B, H, W))
        z = centers[pick].permute(0, 3, 1, 2) * 3.0 + 0.05 * torch.randn(B, C, H, W)
    z.requires_grad_(True)
    zq, usage, vq, commit, _ = q(z)
    g_out = torch.randn_like(zq) * 0.1
    g_vq, g_commit = 1.7, 0.6
    (zq * g_out).sum().add(vq * g_vq).add(commit * g_commit).backward()
    with torch.no_grad():
        idx = q.f_to_idxBl_or_fhat(z.detach(), to_fhat=False, v_patch_nums=None)[0]
        fhat = q.f_to_idxBl_or_fhat(z.detach(), to_fhat=True, v_patch_nums=None)[0]
    np.savez(os.path.join(OUT, name + ".npz"),
             z=z.detach().numpy(), E=q.embedding.weight.detach().numpy(), beta=np.float32(beta),
             codebook_norm=np.int32(codebook_norm), idx=idx.numpy(), zq=zq.detach().numpy(), fhat=fhat.numpy(),
             vq_loss=np.float32(vq.item()), commit_loss=np.float32(commit.item()), usage=np.float32(usage[0]),
             ema_hit=q.ema_vocab_hit_SV.numpy(), g_out=g_out.numpy(), g_vq=np.float32(g_vq),
             g_commit=np.float32(g_commit), g_z=z.grad.numpy(), g_E=q.embedding.weight.grad.numpy(),
             meta=np.array(str(meta())))
    print("wrote", name, "usage", usage[0], "vq", vq.item())


def main():
    os.makedirs(OUT, exist_ok=True)
    # BASELINE config 1 shape (VQ-4096, C=64, B=4, 16x16) — the reference's own CPU-runnable case
    gen_vq("vq_cfg1_v4096_c64_b4", 4096, 64, 4, 16, 16, seed=0)
    # config 2 codebook geometry at a fixture-sized batch (VQ-8192, C=32)
    gen_vq("vq_cfg2_v8192_c32_b2", 8192, 32, 2, 16, 16, seed=1)
    # ragged: N not a multiple of the 32-token tile, V not a multiple of the 128-code stage, tiny C
    gen_vq("vq_ragged_v1000_c8_b3_5x7", 1000, 8, 3, 5, 7, seed=2)
    # no codebook norm (raw L2 path), C=16
    gen_vq("vq_raw_v512_c16_b2", 512, 16, 2, 8, 8, seed=3, codebook_norm=False, z_scale=0.02)
    # clustered latents: many near-duplicates -> histogram contention + near ties
    gen_vq("vq_clustered_v2048_c32_b2", 2048, 32, 2, 16, 16, seed=4, clustered=True)


if __name__ == "__main__":
    main()
