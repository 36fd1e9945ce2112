"""xq_conv3x3_to3_forward (NHWC bf16 C -> 3 planar channels: conv_out of the CNN tokenizer, the data gradient of VGG conv1_1) at the train steps'
shapes: time per launch and a hash of the output, so that XQ_TO3_TILED=0 / 1 (row segments through L1 / L2, 2-D tiles with the halo in LDS) can be
compared across two processes — the two forms are meant to be bit-identical.

    XQ_TO3_TILED=1 python tools/bench_to3.py
"""
import hashlib
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from imagefolder_amd import ops_dense as od  # noqa: E402

tag = f"XQ_TO3_TILED={os.environ.get('XQ_TO3_TILED', '(default)')}"
for C, B, H, W in ((64, 128, 256, 256), (128, 32, 256, 256), (64, 3, 37, 45), (128, 2, 5, 3)):
    torch.manual_seed(C + H)
    x = torch.randn(B, C, H, W, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
    w = torch.randn(3, C, 3, 3, device="cuda") * 0.05
    b = torch.randn(3, device="cuda")
    wq = w.permute(0, 2, 3, 1).reshape(3, 9, C).to(torch.bfloat16).contiguous()
    y = od._to3(x, wq, b)
    torch.cuda.synchronize()
    h = hashlib.sha1(y.view(torch.int16).cpu().numpy().tobytes()).hexdigest()[:16]
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            od._to3(x, wq, b)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 5)
    ms = statistics.median(ts)
    print(f"{tag} C{C} B{B} {H}x{W}: {ms:.4f} ms  {B * H * W * C * 2 / ms / 1e6:7.0f} GB/s of input  sha1 {h}", flush=True)
