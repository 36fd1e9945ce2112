"""Mirror of reference tokenizer/tokenizer_image/latent_perturbation.py (RobustTok latent perturbation).

Same signature and semantics as the reference `add_perturbation`; the distance rows, rank selection,
gather/renormalise/straight-through run in libxq_ops.so (xq_perturb_forward/backward).  Differences
(documented in DESIGN.md): only the first int(B*beta) samples are processed (the reference computes the
N x V matrix and a full top-delta for every token and discards >= 90 % of it), and only the rank that is
actually picked is selected (no top-delta list is materialised).
"""
import torch

from . import ops


def draw_ranks(n_tokens: int, alpha: float, delta: int, device, generator=None):
    """The reference's RNG draws, in the reference's order (latent_perturbation.py:21-23):
    random_prob = rand(N); random_idx = randint(0, delta, (N,)); rank = where(random_prob > alpha, 0, random_idx)."""
    random_prob = torch.rand(n_tokens, device=device, generator=generator)
    random_idx = torch.randint(0, delta, random_prob.shape, device=device, generator=generator)
    return torch.where(random_prob > alpha, torch.zeros_like(random_idx), random_idx)


def add_perturbation(z, z_q, z_channels, codebook_norm, codebook, alpha, beta, delta, rank=None):
    """z: encoder latent h (B,C,H,W); z_q: quantizer output (B,C,H,W); codebook: nn.Embedding.
    `rank` (optional, int tensor (B*H*W,)) overrides the RNG draws — used by the parity tests."""
    B = z.shape[0]
    n_tokens = z.numel() // z_channels
    if rank is None:
        # always drawn, like the reference, so that the device RNG stream stays in step with it
        rank = draw_ranks(n_tokens, alpha, int(delta), z.device)
    n_pert = int(B * beta)  # latent_perturbation.py:32
    return ops.PerturbStraightThrough.apply(z, z_q, codebook.weight, bool(codebook_norm), n_pert, rank)
