"""Dev tool (GPU box): conv3x3 implicit-GEMM kernel vs the library conv at the VGG16 / CNN layer shapes (bf16 NHWC)."""
import sys, os, time, torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imagefolder_amd
from imagefolder_amd.ops_dense import Conv3x3Fn
def bench(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / n
shapes = [(64, 64, 64, 256), (64, 64, 128, 128), (64, 128, 128, 128), (64, 128, 256, 64), (64, 256, 256, 64), (64, 256, 512, 32), (64, 512, 512, 32), (64, 512, 512, 16), (16, 128, 128, 256)]
for B, Cin, Cout, HW in shapes:
    x = torch.randn(B, Cin, HW, HW, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05
    wb = w.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    b = torch.zeros(Cout, device="cuda")
    fl = 2.0 * B * HW * HW * Cin * Cout * 9
    with torch.no_grad():
        t1 = bench(lambda: Conv3x3Fn.apply(x, w, b, True))
        t2 = bench(lambda: torch.relu(F.conv2d(x, wb, b.to(torch.bfloat16), padding=1)))
    print(f"B{B} {Cin}->{Cout} @{HW}: xq {t1*1e3:7.3f} ms {fl/t1/1e12:6.1f} TF/s | miopen {t2*1e3:7.3f} ms {fl/t2/1e12:6.1f} TF/s", flush=True)

# weight gradient: hand-written transpose-read kernel vs the library wgrad (CNN encoder/decoder shapes)
from imagefolder_amd.ops_dense import conv3x3_weight_grad
if "--wgrad" in sys.argv:
    for B, Cin, Cout, HW in [(8, 128, 128, 256), (8, 128, 256, 128), (8, 256, 256, 128), (16, 256, 256, 64), (16, 512, 512, 32), (16, 512, 512, 16)]:
        x = torch.randn(B, Cin, HW, HW, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        g = torch.randn(B, Cout, HW, HW, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        w = torch.randn(Cout, Cin, 3, 3, device="cuda") * 0.05
        wb = w.to(torch.bfloat16)
        fl = 2.0 * B * HW * HW * Cin * Cout * 9
        t1 = bench(lambda: conv3x3_weight_grad(x, g, w))
        t2 = bench(lambda: torch.ops.aten.convolution_backward(g, x, wb, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False]))
        print(f"wgrad B{B} {Cin}->{Cout} @{HW}: xq {t1*1e3:7.3f} ms {fl/t1/1e12:6.1f} TF/s | miopen {t2*1e3:7.3f} ms {fl/t2/1e12:6.1f} TF/s", flush=True)
