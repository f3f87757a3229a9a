"""Time the FLF gate at the C2 latent size with both motion backends.  python tools/flow_bench.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from worldforge_amd.flf import VideoMotionPCASelector

dev = "cuda:0"
g = torch.Generator(device=dev).manual_seed(0)
enc = torch.randn(1, 16, 21, 60, 104, generator=g, device=dev)
pred = (enc + 0.3 * torch.randn(1, 16, 21, 60, 104, generator=g, device=dev)).to(torch.bfloat16)
for backend in ("tdiff", "farneback"):
    sel = VideoMotionPCASelector(flow_backend=backend)
    for _ in range(3):
        sel.select_motion_related_channels(pred, enc, current_step=20)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    n = 10
    for _ in range(n):
        ch = sel.select_motion_related_channels(pred, enc, current_step=20)
    e.record()
    torch.cuda.synchronize()
    print(f"FLF gate, {backend}: {s.elapsed_time(e) / n:.3f} ms per call (incl. the one D2H sync), channels {ch}")
