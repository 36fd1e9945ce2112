"""Time the packed-qkv attention kernels (csrc/xq_attn.hip) against the library SDPA on the step's shapes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from imagefolder_amd import ops_dense


def timeit(fn, n=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    dev = "cuda"
    for (B, N, H) in [(128, 513, 12), (128, 257, 12), (128, 197, 6), (256, 197, 6)]:
        qkv = torch.randn(B, N, 3 * H * 64, device=dev).to(torch.bfloat16).requires_grad_(True)
        g = torch.randn(B, N, H * 64, device=dev).to(torch.bfloat16)
        unit = 2.0 * B * H * N * N * 64  # one N x N x 64 product

        def lib_fwd():
            q, k, v = qkv.view(B, N, 3, H, 64).permute(2, 0, 3, 1, 4).unbind(0)
            return F.scaled_dot_product_attention(q, k, v).transpose(1, 2).reshape(B, N, H * 64)

        def mine_fwd():
            return ops_dense.AttentionFn.apply(qkv, H)

        o_l = lib_fwd()
        o_m = mine_fwd()
        t_lf = timeit(lambda: lib_fwd())
        t_mf = timeit(lambda: mine_fwd())
        t_lb = timeit(lambda: torch.autograd.grad(o_l, qkv, g, retain_graph=True))
        t_mb = timeit(lambda: torch.autograd.grad(o_m, qkv, g, retain_graph=True))
        print(f"B={B} N={N} H={H}: fwd lib {t_lf*1e3:.0f} us ({2*unit/t_lf/1e9:.0f} TF/s) | hip {t_mf*1e3:.0f} us ({2*unit/t_mf/1e9:.0f} TF/s)"
              f" || bwd lib {t_lb*1e3:.0f} us | hip {t_mb*1e3:.0f} us ({7*unit/t_mb/1e9:.0f} TF/s on 7 products, {5*unit/t_mb/1e9:.0f} on 5)",
              flush=True)


if __name__ == "__main__":
    main()
