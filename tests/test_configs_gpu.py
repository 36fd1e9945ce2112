"""GPU: every BASELINE.json config geometry runs one full train step (generator + VQLoss + discriminator step +
fused optimizer) at a reduced batch, with finite losses and updated weights."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,batch", [("VQ-8192", 8), ("VQ-4096-cnn", 2), ("VP2-16384", 10), ("MSVR10P2-4096", 10), ("RobustTok", 10),
                                        ("MSBR10P2-4096", 10)])
def test_one_full_train_step(name, batch, monkeypatch):
    import bench
    from imagefolder_amd import nn_ops
    # assert-no-library mode: every dense op of the bf16 step of EVERY config runs on a hand-written kernel, or this test fails loudly
    monkeypatch.setattr(nn_ops, "STRICT_HIP", True)

    class A:
        pass
    a = A()
    a.batch, a.loss = batch, "full"
    bench.CFG.update(bench.CONFIGS[name])
    dev = torch.device("cuda:0")
    model, ts = bench.build_train_step(a, dev, 1)
    imgs = torch.rand(batch, 3, 256, 256, device=dev) * 2 - 1
    w0 = model.decoder.last_layer.detach().clone()
    for _ in range(2):
        loss = ts.step(imgs, epoch=0, alpha=bench.CFG["alpha"], beta=bench.CFG["beta_lp"], delta=bench.CFG["delta"])
    assert torch.isfinite(loss)
    assert not torch.equal(w0, model.decoder.last_layer.detach())
    assert all(torch.isfinite(p).all() for p in model.parameters())
    model.eval()
    with torch.no_grad():
        rec = model.img_to_reconstructed_img(imgs[:2])
    assert rec.shape == (2, 3, 256, 256)
    bench.CFG.update(bench.CONFIGS["VQ-8192"])
