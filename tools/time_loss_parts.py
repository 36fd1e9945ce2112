"""Dev tool (GPU box): LPIPS and DinoDisc fwd/bwd wall times in isolation (B=128, bf16 autocast)."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import imagefolder_amd
from imagefolder_amd.vq_loss import LPIPS, DinoDisc, DiffAug
from imagefolder_amd import nn_ops, ops_dense
torch.manual_seed(0)
dev = "cuda"
lp = LPIPS().to(dev).eval()
dd = DinoDisc().to(dev).train()
imgs = torch.rand(128, 3, 256, 256, device=dev) * 2 - 1
rec = (torch.rand(128, 3, 256, 256, device=dev) * 2 - 1).requires_grad_(True)
def sync(): torch.cuda.synchronize(); return time.perf_counter()
def run(name, fn, n=4):
    for k in range(n):
        t0 = sync()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = fn()
        t1 = sync(); y.backward(); t2 = sync()
    print(f"{name}: fwd {1e3*(t1-t0):.1f} bwd {1e3*(t2-t1):.1f} ms", flush=True)
mode = sys.argv[1] if len(sys.argv) > 1 else "hip"
if mode == "aten":
    ops_dense.conv3x3_supported = lambda *a, **k: False
run(f"lpips[{mode}]", lambda: lp(imgs, rec).mean())
run("disc", lambda: dd(rec).mean())
run("daug", lambda: DiffAug(1.0, 0.2).aug(rec, 0).mean())
