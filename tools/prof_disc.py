import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagefolder_amd.vq_loss import DinoDisc
torch.manual_seed(0)
d = DinoDisc().cuda().train()
x = (torch.rand(128, 3, 256, 256, device="cuda") * 2 - 1)
def sync(): torch.cuda.synchronize(); return time.perf_counter()
with torch.autocast("cuda", dtype=torch.bfloat16):
    acts = d.dino_proxy[0](x.float())
h = d.heads[0]
a = acts[0]
print("act", a.shape, a.dtype, a.is_contiguous())
for rep in range(3):
    with torch.autocast("cuda", dtype=torch.bfloat16):
        cur = a
        for name, mod in [("blk1.conv", h[0][0]), ("blk1.bn", h[0][1]), ("blk1.act", h[0][2]), ("res.conv9", h[1].fn[0]), ("res.bn", h[1].fn[1]), ("res.act", h[1].fn[2]), ("final conv", h[2])]:
            t0 = sync(); cur2 = mod(cur.clone() if name.endswith("act") else cur); t1 = sync()
            if rep == 2: print(f"  {name:12s} {1e3*(t1-t0):8.2f} ms  out {tuple(cur2.shape)} {cur2.dtype}")
            cur = cur2
