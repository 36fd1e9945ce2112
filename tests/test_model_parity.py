"""Model-level parity with the reference CPU path (BASELINE.json: indices bit-exact, pixels within 1e-4 fp32).

Goldens: tests/golden/model_*.npz = the reference VQModel (deterministic weights, oracle/det_init.py) run on CPU in
fp32: input image, latent f = quant_conv(encoder(x)), code indices, reconstruction.
  * CPU test: the mirror's encoder (library ops) reproduces the reference latent f;
  * GPU test: the whole MI355X path in fp32 reproduces indices and pixels — on hand-written kernels end to end: the HIP
    quantizer, the fused ViT row kernels and the fp32-MFMA convolution / Linear / attention / GroupNorm kernels of
    csrc/xq_f32.hip (the test asserts that no dense op fell back to a library call)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle.det_init import det_state_dict

CASES = {
    "model_cfg1_cnn_vq4096": dict(codebook_size=4096, codebook_embed_dim=64, v_patch_nums=[16], enc_type="cnn", dec_type="cnn",
                                  semantic_guide="none", detail_guide="none", num_latent_tokens=256, product_quant=1),
    "model_cfg2_vitb_vq8192": dict(codebook_size=8192, codebook_embed_dim=32, v_patch_nums=[16], enc_type="dinov2",
                                   dec_type="dinov2", semantic_guide="none", detail_guide="none", num_latent_tokens=256,
                                   product_quant=1, abs_pos_embed=True, encoder_model="vit_base_patch14_dinov2.lvd142m",
                                   decoder_model="vit_base_patch14_dinov2.lvd142m"),
}


def build(name):
    from imagefolder_amd.xqgan_model import VQ_models
    g = load_golden(name)
    torch.manual_seed(0)
    m = VQ_models["VQ-16"](**CASES[name]).eval()
    m.load_state_dict(det_state_dict(m.state_dict(), int(g["seed"])))
    return m, g


@pytest.mark.parametrize("name", sorted(CASES))
def test_cpu_mirror_encoder_latent_matches_reference(name):
    m, g = build(name)
    with torch.no_grad():
        f = m.encode(torch.from_numpy(g["x"]))
    assert np.abs(f.numpy() - g["f"]).max() <= 1e-5 * max(1.0, np.abs(g["f"]).max())


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_gpu_fp32_indices_and_pixels_match_reference_cpu(oracle, name):
    from imagefolder_amd import nn_ops
    m, g = build(name)
    m = m.cuda()
    x = torch.from_numpy(g["x"]).cuda()
    for key in list(nn_ops.IMPL):
        if key.endswith("_fp32_inference") or key == "linear_fp32_training":
            del nn_ops.IMPL[key]
    import torch.nn.functional as F
    lib_calls = []
    saved = {n: getattr(F, n) for n in ("conv2d", "linear", "group_norm", "scaled_dot_product_attention")}

    def spy(n):
        def f(*a, **k):
            t = a[0]
            if isinstance(t, torch.Tensor) and t.is_cuda and t.numel() > 4096:
                lib_calls.append(n)
            return saved[n](*a, **k)
        return f
    for n in saved:
        setattr(F, n, spy(n))
    # ... and the raw products (round 4: ops_dense.LinearFn's fp32 branch called torch.addmm for parameters that require a gradient —
    # autograd reports needs_input_grad for them under no_grad too — and the functional spies above never saw it)
    saved_t = {n: getattr(torch, n) for n in ("addmm", "mm", "bmm", "matmul", "baddbmm")}

    def spy_t(n):
        def f(*a, **k):
            if any(isinstance(t, torch.Tensor) and t.is_cuda and t.numel() > 4096 for t in a):
                lib_calls.append("torch." + n)
            return saved_t[n](*a, **k)
        return f
    for n in saved_t:
        setattr(torch, n, spy_t(n))
    try:
        with torch.no_grad():
            f = m.encode(x)
            idx = m.img_to_idx(x)[0][0].cpu().numpy()
            rec = m.img_to_reconstructed_img(x).cpu().numpy()
    finally:
        for n, fn in saved.items():
            setattr(F, n, fn)
        for n, fn in saved_t.items():
            setattr(torch, n, fn)
    # every dense op of the fp32 path ran on a hand-written kernel
    assert not lib_calls, f"library ops on the fp32 parity path: {sorted(set(lib_calls))}"
    want = ["conv2d_fp32_inference", "group_norm_silu_fp32_inference", "spatial_attention_fp32_inference"] if CASES[name]["enc_type"] == "cnn" \
        else ["linear_fp32_inference", "attention_fp32_inference"]   # (the blocks' Linear layers: LinearFn's fp32 inference branch, grad mode off)
    for key in want:
        assert nn_ops.IMPL.get(key, "").startswith("hip"), (key, nn_ops.IMPL.get(key))
    # latent within fp32 rounding of the CPU reference
    assert np.abs(f.cpu().numpy() - g["f"]).max() <= 2e-4 * max(1.0, np.abs(g["f"]).max())
    E = m.quantize.embedding.weight.detach().cpu().numpy()
    # indices: exact, except tokens whose two candidate codes are an fp64-verified near tie ON THE REFERENCE latent
    # (measured, rounds 1-3: 0 mismatches on both configs; the slack is two tokens, each of which must be such a tie)
    par = oracle.index_parity(g["f"], E, oracle.MODE_L2_NORMED, idx, g["idx"], tol=2e-4)
    assert par["n_mismatch"] <= 2 and par["all_ties"], par
    # the HIP quantizer on the REFERENCE latent is bit-exact with the reference's indices
    from imagefolder_amd import ops
    idx_ref_latent = ops.assign(torch.from_numpy(g["f"]).cuda(), m.quantize.embedding.weight, ops.MODE_L2_NORMED).cpu().numpy()
    par2 = oracle.index_parity(g["f"], E, oracle.MODE_L2_NORMED, idx_ref_latent, g["idx"])
    assert par2["n_mismatch"] == 0 or par2["all_ties"], par2
    # pixels: <= 1e-4 wherever the 16x16-pixel patch's token (and, for the CNN, its receptive field) kept its code
    if par["n_mismatch"] == 0:
        assert np.abs(rec - g["rec"]).max() <= 1e-4
    else:  # a flipped token (an fp64-verified tie, at most two) changes its neighbourhood legitimately; the rest must still agree
        assert np.mean(np.abs(rec - g["rec"]) <= 1e-4) >= 0.97


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_gpu_bf16_kernels_are_as_close_to_fp32_as_the_reference_bf16_path(name):
    """The TRAINING kernels (bf16 MFMA convolution / GEMM / attention, bf16 GroupNorm) on the reference-parity path.
    bf16 results cannot meet the 1e-4 fp32 bound; the bound is DERIVED: tests/golden/*_bf16.npz holds what the unmodified reference
    produces for the same image under torch.autocast(bfloat16) on CPU (oracle/make_golden.py gen_model_bf16), and its distance to
    the reference's own fp32 result is the yardstick — the MI355X bf16 path must be at most 1.5x as far from the fp32 golden
    (rms over the latent and over the pixels) and may flip at most twice as many code indices (+2)."""
    from imagefolder_amd import nn_ops
    m, g = build(name)
    gb = load_golden(name + "_bf16")
    m = m.cuda()
    x = torch.from_numpy(g["x"]).cuda()
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        f = m.encode(x).float().cpu().numpy()
        idx = m.img_to_idx(x)[0][0].cpu().numpy()
        rec = m.img_to_reconstructed_img(x).float().cpu().numpy()
    if CASES[name]["enc_type"] == "cnn":
        assert nn_ops.IMPL["conv2d"].startswith("hip") and nn_ops.IMPL["group_norm_silu"] == "hip"
    else:
        assert nn_ops.IMPL["attention"] == "hip" and nn_ops.IMPL["linear"].startswith("hip")

    def rms(a):
        return float(np.sqrt(np.mean(np.square(a, dtype=np.float64))))
    ref_f, ref_rec = rms(gb["f_bf16"] - g["f"]), rms(gb["rec_bf16"] - g["rec"])
    ref_flips = int((gb["idx_bf16"] != g["idx"]).sum())
    our_f, our_rec, our_flips = rms(f - g["f"]), rms(rec - g["rec"]), int((idx != g["idx"]).sum())
    print(f"{name}: latent rms {our_f:.4e} (reference bf16 {ref_f:.4e}); pixel rms {our_rec:.4e} (reference bf16 {ref_rec:.4e}); "
          f"code flips {our_flips} (reference bf16 {ref_flips}) of {idx.size}")
    assert our_f <= 1.5 * ref_f
    assert our_rec <= 1.5 * ref_rec
    assert our_flips <= 2 * ref_flips + 2
