"""Dev tool (GPU box): wall-time split of the full train step (synchronised sections)."""
import sys, time, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
class A: batch=128; loss="full"
dev = torch.device("cuda:0")
model, ts = bench.build_train_step(A, dev, 1)
imgs = torch.rand(128, 3, 256, 256, device=dev) * 2 - 1
def sync(): torch.cuda.synchronize(); return time.perf_counter()
for it in range(5):
    ts.arena.rebind_grads()
    t0 = sync()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = ts.model(imgs, 0, 0.0, 0.0, 100)
    t1 = sync()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss = ts.gen_loss_fn(out, imgs)
    t2 = sync()
    loss.backward()
    t3 = sync()
    ts.disc_step_fn(imgs, out[0].detach())
    t4 = sync()
    ts.opt.step()
    t5 = sync()
    print(f"it{it}: model fwd {1e3*(t1-t0):.1f} | vq_loss(gen) {1e3*(t2-t1):.1f} | backward {1e3*(t3-t2):.1f} | disc step {1e3*(t4-t3):.1f} | opt {1e3*(t5-t4):.1f} ms", flush=True)
# finer: inside vq_loss gen
vl = ts.disc_step_fn.vq_loss
rec = out[0].detach().requires_grad_(True)
for name, fn in [("lpips", lambda: vl.perceptual_loss(imgs, rec).mean()), ("daug", lambda: vl.daug.aug(rec, 0).mean()), ("disc", lambda: vl.discriminator(rec).mean())]:
    for k in range(3):
        t0 = sync()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            y = fn()
        t1 = sync(); y.backward(); t2 = sync()
    print(f"{name}: fwd {1e3*(t1-t0):.1f} bwd {1e3*(t2-t1):.1f} ms")
