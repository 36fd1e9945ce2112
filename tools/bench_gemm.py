"""Dev tool (GPU box): library GEMM rates for the ViT-B train-step shapes, incl. split-K variants of the weight-gradient GEMM."""
import torch, time, sys
dev = "cuda"
M = 128 * 513
def bench(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
shapes = [("qkv", 2304, 768), ("proj", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)]
for name, N, K in shapes:
    x = torch.randn(M, K, device=dev, dtype=torch.bfloat16)
    W = torch.randn(N, K, device=dev, dtype=torch.bfloat16)
    b = torch.randn(N, device=dev, dtype=torch.bfloat16)
    g = torch.randn(M, N, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * M * N * K
    t = bench(lambda: torch.addmm(b, x, W.t())); print(f"{name:5s} fwd  addmm          {t*1e3:7.3f} ms {fl/t/1e12:7.1f} TF/s")
    t = bench(lambda: torch.mm(g, W)); print(f"{name:5s} gx   mm             {t*1e3:7.3f} ms {fl/t/1e12:7.1f} TF/s")
    t = bench(lambda: torch.mm(g.t(), x)); print(f"{name:5s} gW   mm bf16out    {t*1e3:7.3f} ms {fl/t/1e12:7.1f} TF/s")
    t = bench(lambda: torch.mm(g.t(), x, out_dtype=torch.float32)); print(f"{name:5s} gW   mm f32out     {t*1e3:7.3f} ms {fl/t/1e12:7.1f} TF/s")
    for S in (4, 8, 16, 32):
        gs, xs = g.view(S, M // S, N), x.view(S, M // S, K)
        t = bench(lambda: torch.bmm(gs.transpose(1, 2), xs, out_dtype=torch.float32).sum(0)); print(f"{name:5s} gW   bmm S={S:2d} f32    {t*1e3:7.3f} ms {fl/t/1e12:7.1f} TF/s")
    # gW as x^T g then transposed view (other operand order)
    t = bench(lambda: torch.mm(x.t(), g, out_dtype=torch.float32)); print(f"{name:5s} gW^T mm f32out     {t*1e3:7.3f} ms {fl/t/1e12:7.1f} TF/s")
