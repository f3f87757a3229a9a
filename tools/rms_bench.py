"""HBM rate of the fused RMS_norm + SiLU kernel at the VAE's full-resolution shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from worldforge_amd import _ffi, ops
for (n, C) in ((81 * 480 * 832, 96), (81 * 240 * 416, 192), (41 * 120 * 208, 384)):
    x = torch.randn(n, C, device="cuda:0")
    g = torch.ones(C, device="cuda:0")
    ob = torch.empty(n, C, device="cuda:0", dtype=torch.bfloat16)
    run = lambda: _ffi.call("wf_rms_silu_cl", x.data_ptr(), g.data_ptr(), ob.data_ptr(), None, n, C, 1, ops.stream())
    run(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): run()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 5
    print(f"rms_silu n={n} C={C}: {ms:.3f} ms  {n * C * 6 / ms / 1e9:.2f} TB/s (f32 in, bf16 out)")
