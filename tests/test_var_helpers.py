"""VAR-side helpers of VectorQuantizer2: embed_to_fhat, idxBl_to_var_input, get_next_autoregressive_input
(reference tokenizer/tokenizer_image/quant.py:148-180, :226-258 and the identical methods of models/quant.py).

  * CPU: the oracle's restatement on its C ladder primitives vs goldens recorded from the reference methods
    (oracle/make_golden.py gen_var_helpers);
  * GPU: the mirror's methods (ops.ms_upsample / ms_phi_accumulate / ms_area_pool = the ladder's own kernels through the
    C-ABI) bit-identical to the oracle, and within fp32 round-off of the reference goldens.
Tolerance vs the reference: 3e-5 absolute on O(1) values — ATen evaluates the bicubic taps in fp32 (3e-6 off fp64 per
interpolation, tests/test_oracle_golden.py), and up to 10 scales accumulate."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import xq_oracle

GOLD = os.path.join(os.path.dirname(__file__), "golden")
FIXTURES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLD, "var_helpers_*.npz")))
assert len(FIXTURES) == 3
LFQ_FIXTURES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLD, "vh_lfq_*.npz")))
assert len(LFQ_FIXTURES) == 2


def _load(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    pns = [int(p) for p in g["pns"]]
    B = g["fhat_last"].shape[0]
    idx, off = [], 0
    for pn in pns:
        idx.append(g["idx"][off:off + B * pn * pn].reshape(B, pn * pn))
        off += B * pn * pn
    ms_h = [xq_oracle.gather_nchw(g["E"], i, pn) for i, pn in zip(idx, pns)]
    nexts, off = [], 0
    C = g["E"].shape[1]
    for si, pn in enumerate(pns):
        p2 = pns[si + 1] if si + 1 < len(pns) else pns[-1]
        n = B * C * p2 * p2
        nexts.append(g["next_maps"][off:off + n].reshape(B, C, p2, p2))
        off += n
    return g, pns, idx, ms_h, nexts


@pytest.mark.parametrize("name", FIXTURES)
def test_oracle_var_helpers_vs_reference_golden(name):
    g, pns, idx, ms_h, nexts = _load(name)
    sel, pw, pb, r = g["phi_sel"], g["phi_w"], g["phi_b"], float(g["phi_ratio"])
    fh = xq_oracle.embed_to_fhat(ms_h, pns, sel, pw, pb, r)
    np.testing.assert_allclose(np.stack(fh), g["fhat_scales"], atol=3e-5, rtol=0)
    np.testing.assert_allclose(fh[-1], g["fhat_last"], atol=3e-5, rtol=0)
    vi = xq_oracle.idxBl_to_var_input(idx, g["E"], pns, sel, pw, pb, r)
    assert vi.shape == g["var_input"].shape
    np.testing.assert_allclose(vi, g["var_input"], atol=3e-5, rtol=0)
    f_hat = np.zeros_like(g["fhat_last"])
    for si in range(len(pns)):
        f_hat, nxt = xq_oracle.next_autoregressive_input(si, f_hat, ms_h[si], pns, sel, pw, pb, r)
        np.testing.assert_allclose(nxt, nexts[si], atol=3e-5, rtol=0)
    np.testing.assert_allclose(f_hat, g["f_hat_final"], atol=3e-5, rtol=0)


def _mirror(g, pns, dev):
    from imagefolder_amd.quant import VectorQuantizer2, VectorQuantizer2Var
    V, C = g["E"].shape
    share = int(g["share"])
    if int(g["var_variant"]):
        q = VectorQuantizer2Var(V, C, True, v_patch_nums=tuple(pns), share_quant_resi=share)
    else:
        q = VectorQuantizer2(V, C, using_znorm=True, v_patch_nums=list(pns), num_latent_tokens=pns[-1] ** 2, share_quant_resi=share)
    q.embedding.weight.data.copy_(torch.from_numpy(g["E"]))
    for k, c in enumerate(q.quant_resi.convs()):
        c.weight.data.copy_(torch.from_numpy(g["phi_w"][k]))
        c.bias.data.copy_(torch.from_numpy(g["phi_b"][k]))
    return q.to(dev).eval()


@pytest.mark.gpu
@pytest.mark.parametrize("name", FIXTURES)
def test_hip_var_helpers_equal_oracle_and_reference(name):
    g, pns, idx, ms_h, nexts = _load(name)
    dev = torch.device("cuda")
    q = _mirror(g, pns, dev)
    sel, pw, pb, r = g["phi_sel"], g["phi_w"], g["phi_b"], float(g["phi_ratio"])
    t_idx = [torch.from_numpy(i).to(dev) for i in idx]
    t_h = [torch.from_numpy(h).to(dev) for h in ms_h]
    with torch.no_grad():
        fh = q.embed_to_fhat(t_h, all_to_max_scale=True, last_one=False)
        fh_last = q.embed_to_fhat(t_h, all_to_max_scale=True, last_one=True)
        vi = q.idxBl_to_var_input(t_idx)
        o_fh = xq_oracle.embed_to_fhat(ms_h, pns, sel, pw, pb, r)
        for a, b in zip(fh, o_fh):
            assert np.array_equal(a.cpu().numpy(), b), "embed_to_fhat differs from the oracle"
        assert np.array_equal(fh_last.cpu().numpy(), o_fh[-1])
        assert np.array_equal(vi.cpu().numpy(), xq_oracle.idxBl_to_var_input(idx, g["E"], pns, sel, pw, pb, r))
        np.testing.assert_allclose(vi.cpu().numpy(), g["var_input"], atol=3e-5, rtol=0)
        np.testing.assert_allclose(torch.stack(fh).cpu().numpy(), g["fhat_scales"], atol=3e-5, rtol=0)
        f_hat = torch.zeros_like(t_h[-1])
        o_f = np.zeros_like(g["fhat_last"])
        for si in range(len(pns)):
            keep = f_hat
            f_hat, nxt = q.get_next_autoregressive_input(si, len(pns), f_hat, t_h[si])
            assert f_hat.data_ptr() == keep.data_ptr(), "f_hat must be updated in place (quant.py:253)"
            o_f, o_n = xq_oracle.next_autoregressive_input(si, o_f, ms_h[si], pns, sel, pw, pb, r)
            assert np.array_equal(nxt.cpu().numpy(), o_n)
            np.testing.assert_allclose(nxt.cpu().numpy(), nexts[si], atol=3e-5, rtol=0)
        np.testing.assert_allclose(f_hat.cpu().numpy(), g["f_hat_final"], atol=3e-5, rtol=0)


@pytest.mark.gpu
def test_model_level_var_wrappers_split_the_product_branches():
    """VQModel.get_next_autoregressive_input / idxBl_to_var_input (xqgan_model.py:434-451) with product_quant = 2: channel
    chunks are non-contiguous views and must still be updated in place."""
    from imagefolder_amd.quant import VectorQuantizer2
    dev = torch.device("cuda")
    pns = [1, 2, 3, 4]
    torch.manual_seed(0)
    qs = [VectorQuantizer2(64, 8, using_znorm=True, v_patch_nums=pns, num_latent_tokens=16).to(dev).eval() for _ in range(2)]
    B = 2
    f_hat = torch.zeros(B, 16, 4, 4, device=dev)
    h = torch.randn(B, 16, 2, 2, device=dev)
    outs = []
    for i, (fc, hc) in enumerate(zip(f_hat.chunk(2, dim=1), h.chunk(2, dim=1))):
        ref = qs[i].quant_resi[1 / 3](torch.nn.functional.interpolate(hc, size=(4, 4), mode="bicubic"))
        of, on = qs[i].get_next_autoregressive_input(1, 4, fc, hc)
        outs.append((of, on, ref))
    for i, (of, on, ref) in enumerate(outs):
        torch.testing.assert_close(f_hat[:, 8 * i:8 * i + 8], ref, atol=3e-5, rtol=0)   # written through the view
        torch.testing.assert_close(on, torch.nn.functional.interpolate(ref, size=(3, 3), mode="area"), atol=3e-5, rtol=0)


# ---- LFQ twins (lookup_free_quantize.py:311-343, :404-415): the same two helpers on sign-code maps ------------------------------------
def _load_lfq(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    pns = [int(p) for p in g["pns"]]
    B, C = g["fhat_last"].shape[:2]
    ms_h, off = [], 0
    for pn in pns:
        n = B * C * pn * pn
        ms_h.append(g["ms_h"][off:off + n].reshape(B, C, pn, pn))
        off += n
    nexts, off = [], 0
    for si in range(len(pns)):
        p2 = pns[si + 1] if si + 1 < len(pns) else pns[-1]
        n = B * C * p2 * p2
        nexts.append(g["next_maps"][off:off + n].reshape(B, C, p2, p2))
        off += n
    return g, pns, ms_h, nexts


@pytest.mark.parametrize("name", LFQ_FIXTURES)
def test_oracle_lfq_var_helpers_vs_reference_golden(name):
    g, pns, ms_h, nexts = _load_lfq(name)
    sel, pw, pb, r = g["phi_sel"], g["phi_w"], g["phi_b"], float(g["phi_ratio"])
    fh = xq_oracle.embed_to_fhat(ms_h, pns, sel, pw, pb, r)
    np.testing.assert_allclose(np.stack(fh), g["fhat_scales"], atol=3e-5, rtol=0)
    f_hat = np.zeros_like(g["fhat_last"])
    for si in range(len(pns)):
        f_hat, nxt = xq_oracle.next_autoregressive_input(si, f_hat, ms_h[si], pns, sel, pw, pb, r)
        np.testing.assert_allclose(nxt, nexts[si], atol=3e-5, rtol=0)
    np.testing.assert_allclose(f_hat, g["f_hat_final"], atol=3e-5, rtol=0)


def test_lfq_idxBl_to_var_input_raises_like_upstream():
    """upstream's LFQ.idxBl_to_var_input dereferences self.embedding, which the class does not define (lookup_free_quantize.py:395)"""
    from imagefolder_amd.lookup_free_quantize import LFQ
    q = LFQ(2 ** 8, 8, v_patch_nums=[1, 2, 4], num_latent_tokens=16).eval()
    with pytest.raises(AttributeError):
        q.idxBl_to_var_input([torch.zeros(2, 1, dtype=torch.long), torch.zeros(2, 4, dtype=torch.long)])


@pytest.mark.gpu
@pytest.mark.parametrize("name", LFQ_FIXTURES)
def test_hip_lfq_var_helpers_equal_oracle_and_reference(name):
    from imagefolder_amd.lookup_free_quantize import LFQ
    g, pns, ms_h, nexts = _load_lfq(name)
    dev = torch.device("cuda")
    C = int(g["Cbits"])
    q = LFQ(2 ** C, C, using_znorm=bool(g["using_znorm"]), v_patch_nums=pns, num_latent_tokens=pns[-1] ** 2, share_quant_resi=int(g["share"]),
            codebook_drop=0.0, scale=1.0, entropy_weight=0.0, soft_entropy=True)
    for k, c in enumerate(q.quant_resi.convs()):
        c.weight.data.copy_(torch.from_numpy(g["phi_w"][k]))
        c.bias.data.copy_(torch.from_numpy(g["phi_b"][k]))
    q = q.to(dev).eval()
    sel, pw, pb, r = g["phi_sel"], g["phi_w"], g["phi_b"], float(g["phi_ratio"])
    t_h = [torch.from_numpy(h).to(dev) for h in ms_h]
    with torch.no_grad():
        fh = q.embed_to_fhat(t_h, all_to_max_scale=True, last_one=False)
        o_fh = xq_oracle.embed_to_fhat(ms_h, pns, sel, pw, pb, r)
        for a, b in zip(fh, o_fh):
            assert np.array_equal(a.cpu().numpy(), b), "embed_to_fhat differs from the oracle"
        np.testing.assert_allclose(torch.stack(fh).cpu().numpy(), g["fhat_scales"], atol=3e-5, rtol=0)
        f_hat = torch.zeros_like(t_h[-1])
        o_f = np.zeros_like(g["fhat_last"])
        for si in range(len(pns)):
            f_hat, nxt = q.get_next_autoregressive_input(si, len(pns), f_hat, t_h[si])
            o_f, o_n = xq_oracle.next_autoregressive_input(si, o_f, ms_h[si], pns, sel, pw, pb, r)
            assert np.array_equal(nxt.cpu().numpy(), o_n)
            np.testing.assert_allclose(nxt.cpu().numpy(), nexts[si], atol=3e-5, rtol=0)
        np.testing.assert_allclose(f_hat.cpu().numpy(), g["f_hat_final"], atol=3e-5, rtol=0)
