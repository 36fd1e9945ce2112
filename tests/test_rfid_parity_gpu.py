"""R1 (north_star: "rFID within 0.02 of reference on held-out synthetic batches"): the rFID of MI355X reconstructions equals
the rFID of the reference CPU path's reconstructions of the same synthetic images.

Both sides run the SAME deterministic-weight CNN VQ-16 tokenizer (oracle/det_init.py) on 64 x 64 synthetic images (the CPU
side must finish in about a minute; the layer stack is the full BASELINE config 1 one):
  * reference CPU path = the host mirror (bit-identical to the reference classes on CPU, tests/test_model_parity.py) with
    the nearest-code step from the C oracle (itself pinned to the reference, tests/test_oracle_golden.py);
  * MI355X path = VQModel.img_to_reconstructed_img on the GPU in fp32.
Features: a fixed random conv net standing in for the TF-Inception graph, which is not available offline;
the images are quantised to uint8 first, as xqgan_train.py:526-527 does before the evaluator sees them."""
import numpy as np
import pytest
import torch

from oracle import xq_oracle
from oracle.det_init import det_state_dict

pytestmark = pytest.mark.gpu

KW = dict(codebook_size=4096, codebook_embed_dim=64, v_patch_nums=[16], enc_type="cnn", dec_type="cnn", semantic_guide="none",
          detail_guide="none", num_latent_tokens=256, product_quant=1)


class StandInFeatures(torch.nn.Module):
    def __init__(self, dim=48):
        super().__init__()
        g = torch.Generator().manual_seed(99)
        self.w1 = torch.nn.Parameter(torch.randn(16, 3, 5, 5, generator=g) * 0.2, requires_grad=False)
        self.w2 = torch.nn.Parameter(torch.randn(dim, 16, 3, 3, generator=g) * 0.1, requires_grad=False)

    def forward(self, x255):
        h = torch.nn.functional.conv2d(x255 / 127.5 - 1.0, self.w1, stride=2, padding=2)
        h = torch.nn.functional.conv2d(torch.tanh(h), self.w2, stride=2, padding=1)
        return torch.tanh(h).mean(dim=(2, 3)) * 10.0


def test_rfid_of_gpu_reconstructions_equals_reference_cpu_path():
    from imagefolder_amd import rfid
    from imagefolder_amd.xqgan_model import VQ_models
    torch.manual_seed(0)
    m_cpu = VQ_models["VQ-16"](**KW).eval()
    m_cpu.load_state_dict(det_state_dict(m_cpu.state_dict(), 31))
    m_gpu = VQ_models["VQ-16"](**KW).eval()
    m_gpu.load_state_dict(m_cpu.state_dict())
    m_gpu = m_gpu.cuda()
    feat_cpu, feat_gpu = StandInFeatures(), StandInFeatures().cuda()
    E = m_cpu.quantize.embedding.weight.detach().numpy()

    g = torch.Generator().manual_seed(4321)
    N, bs = 96, 16
    ev_cpu = rfid.ReconstructionFID(feat_cpu, 48)
    ev_gpu = rfid.ReconstructionFID(feat_gpu, 48, device="cuda")
    worst = 0.0
    flips = 0
    with torch.no_grad():
        for _ in range(N // bs):
            # smooth synthetic images: low-frequency random fields in [-1, 1]
            x = torch.nn.functional.interpolate(torch.rand(bs, 3, 8, 8, generator=g) * 2 - 1, size=(64, 64), mode="bicubic").clamp(-1, 1)
            f = m_cpu.encode(x)
            idx, _ = xq_oracle.assign(f.numpy(), E, xq_oracle.MODE_L2_NORMED)
            zq, _, _ = xq_oracle.vq_finish(f.numpy(), E, idx, normed=True, ste=False, want_hist=False)
            rec_cpu = m_cpu.decode(torch.from_numpy(zq)).clamp_(-1, 1)
            xg = x.cuda()
            rec_gpu = m_gpu.img_to_reconstructed_img(xg)
            flips += int((m_gpu.img_to_idx(xg)[0][0].cpu().numpy().reshape(-1) != idx).sum())
            worst = max(worst, (rec_gpu.cpu() - rec_cpu).abs().max().item())
            ev_cpu.update(x, rec_cpu)
            ev_gpu.update(xg, rec_gpu)
    fid_cpu = ev_cpu.compute()
    fid_gpu = ev_gpu.compute()
    fid_gpu_dev = ev_gpu.compute(on_device=True)
    print(f"rFID reference-CPU path {fid_cpu:.6f}  MI355X {fid_gpu:.6f} (on-device eig form {fid_gpu_dev:.6f}); "
          f"max |pixel diff| {worst:.2e}; code flips {flips} of {N * 16}")
    assert fid_cpu > 1e-3, "reconstructions equal the inputs: the test would be vacuous"
    assert abs(fid_gpu - fid_cpu) <= 0.02
    assert abs(fid_gpu_dev - fid_gpu) <= 1e-6 * max(1.0, fid_gpu)
    if flips == 0:
        assert worst <= 1e-4
