"""Does torch's device RNG advance across hipGraph replays on this ROCm build?  (CapturedStep relies on it: DropPath masks, DiffAug
draws, perturbation ranks and the quantizer-dropout depths must differ from replay to replay.)"""
import torch
dev = "cuda"
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    torch.randint(1, 4, (8,), device=dev)
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    r = torch.randint(1, 4, (8,), device=dev)
    u = torch.rand(4, device=dev)
    b = torch.bernoulli(torch.full((6,), 0.5, device=dev))
for i in range(4):
    g.replay()
    torch.cuda.synchronize()
    print("replay", i, r.tolist(), [round(v, 4) for v in u.tolist()], b.tolist())
