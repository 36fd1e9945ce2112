"""nn.Linear's fp32 TRAINING step on the hand-written fp32-MFMA kernels (csrc/xq_f32.hip: conv2d_f32_kernel forward / data gradient,
gemm_f32_tn_kernel weight gradient; ops_f32.LinearF32Fn and the fp32 branch of ops_dense.LinearFn): what leg (a) of
tests/test_train_backward_parity.py trains through.  Against float64 products of the same operands: every output is ONE fp32 fma chain, so
the error bound is the chain's (depth x 2^-24 x sum |terms|), checked element by element."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _chain_bound(abs_terms_sum, depth):
    return abs_terms_sum * depth * 2.0 ** -24 + 1e-30


@pytest.mark.parametrize("M,K,N,bias", [(514, 768, 2304, True), (37, 588, 768, True), (1028, 32, 768, False), (3, 768, 1, True), (130, 100, 70, True)])
@pytest.mark.parametrize("entry", ["ops_f32", "ops_dense", "nn_ops"])
def test_linear_fp32_forward_and_gradients(M, K, N, bias, entry, monkeypatch):
    from imagefolder_amd import nn_ops, ops_dense, ops_f32
    monkeypatch.setattr(nn_ops, "F32_TRAIN_LINEAR", True)      # parity kernels: off by default since round 5
    torch.manual_seed(M + K + N)
    x = torch.randn(2, M // 2 if M % 2 == 0 else M, K, device="cuda") if M % 2 == 0 else torch.randn(M, K, device="cuda")
    x.requires_grad_(True)
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).requires_grad_(True)
    b = torch.randn(N, device="cuda").requires_grad_(True) if bias else None
    nn_ops.IMPL.pop("linear_fp32_training", None)
    if entry == "ops_f32":
        y = ops_f32.LinearF32Fn.apply(x, w, b)
    elif entry == "ops_dense":
        y = ops_dense.LinearFn.apply(x, w, b, False)
    else:
        y = nn_ops.linear(x, w, b)
    if entry != "ops_f32":
        assert nn_ops.IMPL.get("linear_fp32_training", "").startswith("hip")
    g = torch.randn_like(y)
    y.backward(g)
    x64, w64, g64 = x.detach().double().reshape(-1, K), w.detach().double(), g.double().reshape(-1, N)
    y64 = x64 @ w64.t() + (b.detach().double() if bias else 0.0)
    err = (y.detach().double().reshape(-1, N) - y64).abs()
    assert bool((err <= _chain_bound(x64.abs() @ w64.abs().t() + (b.detach().abs().double() if bias else 0.0), K + 1)).all()), float(err.max())
    gx64 = g64 @ w64
    err = (x.grad.double().reshape(-1, K) - gx64).abs()
    assert bool((err <= _chain_bound(g64.abs() @ w64.abs(), N)).all()), float(err.max())
    gw64 = g64.t() @ x64
    err = (w.grad.double() - gw64).abs()
    assert bool((err <= _chain_bound(g64.abs().t() @ x64.abs(), x64.shape[0])).all()), float(err.max())
    if bias:
        err = (b.grad.double() - g64.sum(0)).abs()
        assert bool((err <= _chain_bound(g64.abs().sum(0), x64.shape[0])).all()), float(err.max())


def test_weight_gradient_is_deterministic_and_handles_empty_batches():
    from imagefolder_amd import _lib
    from imagefolder_amd._lib import ptr
    torch.manual_seed(0)
    a, b = torch.randn(1000, 96, device="cuda"), torch.randn(1000, 200, device="cuda")
    outs = []
    for _ in range(3):
        c = torch.full((96, 200), float("nan"), device="cuda")
        assert _lib.lib().xq_gemm_f32_tn(ptr(a), ptr(b), 1000, 96, 200, ptr(c), None) == 0
        outs.append(c)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    c = torch.full((96, 200), float("nan"), device="cuda")
    assert _lib.lib().xq_gemm_f32_tn(None, None, 0, 96, 200, ptr(c), None) == 0
    torch.cuda.synchronize()
    assert bool((c == 0).all())


@pytest.mark.parametrize("B,N,H,hd", [(2, 257, 12, 64), (3, 70, 6, 64), (1, 197, 3, 32), (2, 5, 2, 64)])
def test_attention_fp32_forward_and_gradients(B, N, H, hd, monkeypatch):
    """ops_f32.AttentionF32Fn (attention_f32_kernel + lse, attention_f32_bwd_q / _kv kernels) against float64 softmax attention and its
    autograd on the same packed projection; run twice: bit-identical (no atomics)."""
    from imagefolder_amd import nn_ops, ops_dense, ops_f32
    monkeypatch.setattr(nn_ops, "F32_TRAIN_LINEAR", True)      # parity kernels: off by default since round 5
    torch.manual_seed(B + N + H)
    C = H * hd
    qkv = (torch.randn(B, N, 3 * C, device="cuda") * 0.7).requires_grad_(True)
    assert ops_f32.attention_trainable(qkv, H)
    nn_ops.IMPL.pop("attention_fp32_training", None)
    y = ops_dense.attention_qkvpacked(qkv, H)
    assert nn_ops.IMPL.get("attention_fp32_training", "").startswith("hip")
    g = torch.randn_like(y)
    (dqkv,) = torch.autograd.grad(y, qkv, g)
    q64 = qkv.detach().double().requires_grad_(True)
    q, k, v = q64.view(B, N, 3, H, hd).permute(2, 0, 3, 1, 4).unbind(0)
    p = torch.softmax(q @ k.transpose(-1, -2) * hd ** -0.5, dim=-1)
    y64 = (p @ v).transpose(1, 2).reshape(B, N, C)
    (d64,) = torch.autograd.grad(y64, q64, g.double())
    assert (y.double() - y64).abs().max().item() <= 2e-6 * max(1.0, y64.abs().max().item())
    assert ((dqkv.double() - d64).norm() / d64.norm()).item() <= 2e-6
    assert (dqkv.double() - d64).abs().max().item() <= 1e-5 * d64.abs().max().item()
    y2 = ops_f32.AttentionF32Fn.apply(qkv, H)
    (d2,) = torch.autograd.grad(y2, qkv, g)
    assert torch.equal(y2, y) and torch.equal(d2, dqkv)
