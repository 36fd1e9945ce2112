"""Dev probe (GPU box): can the complete train step be captured into one hipGraph (torch.cuda.CUDAGraph) and replayed?"""
import argparse, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
ap = argparse.ArgumentParser(); ap.add_argument("--config", default="VQ-4096-cnn"); ap.add_argument("--batch", type=int, default=4)
a = ap.parse_args()
bench.CFG.update(bench.CONFIGS[a.config])
args = argparse.Namespace(batch=a.batch, loss="full")
dev = torch.device("cuda:0")
model, ts = bench.build_train_step(args, dev, 1)
imgs = torch.rand(a.batch, 3, 256, 256, device=dev) * 2 - 1
def step(): return ts.step(imgs, epoch=0, alpha=bench.CFG["alpha"], beta=bench.CFG["beta_lp"], delta=bench.CFG["delta"])
def timeit(fn, n=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
print("eager ms/step", timeit(step), flush=True)
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        loss = step()
    print("captured", flush=True)
    print("graph ms/step", timeit(g.replay), "loss", float(loss), flush=True)
except Exception as e:
    import traceback; traceback.print_exc()
    print("CAPTURE FAILED:", type(e).__name__, str(e)[:500])
