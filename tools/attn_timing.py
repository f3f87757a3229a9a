"""Per-phase cycle accounting of the attention main loop (needs a build with WF_EXTRA_HIPCC_FLAGS=-DWF_ATTN_TIMING).
python tools/attn_timing.py"""
import ctypes
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from worldforge_amd import _ffi, dit

L, H = 32760, 40
dev = "cuda:0"
Lp = (L + 63) // 64 * 64
q = torch.randn(H, L, 128, device=dev).to(torch.bfloat16)
k = torch.zeros(H, Lp, 128, device=dev, dtype=torch.bfloat16)
k[:, :L] = torch.randn(H, L, 128, device=dev).to(torch.bfloat16)
vt = torch.randn(H, Lp // 64, 128, 64, device=dev).to(torch.bfloat16)
out = torch.empty(L, H * 128, device=dev, dtype=torch.bfloat16)
lib = _ffi.lib()
buf = (ctypes.c_ulonglong * 16)()
dit.attention(q, k, vt, out, L, 1 / math.sqrt(128))
torch.cuda.synchronize()
lib.wf_debug_attn_cycles(buf, 1)
dit.attention(q, k, vt, out, L, 1 / math.sqrt(128))
torch.cuda.synchronize()
lib.wf_debug_attn_cycles(buf, 1)
names = ["ph0", "wait0", "ph2", "wait2", "ph4", "wait4"]
for g in range(2):
    tiles = buf[g * 8 + 6]
    vals = [buf[g * 8 + i] / max(tiles, 1) for i in range(6)]
    print(f"group {'AB'[g]}: " + "  ".join(f"{n} {v:7.1f}" for n, v in zip(names, vals)) + f"   sum {sum(vals):7.1f} cycles/tile")
