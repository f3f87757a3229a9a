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
buf = (ctypes.c_ulonglong * 64)()
dit.attention(q, k, vt, out, L, 1 / math.sqrt(128))
torch.cuda.synchronize()
lib.wf_debug_attn_cycles(buf, 1)
dit.attention(q, k, vt, out, L, 1 / math.sqrt(128))
torch.cuda.synchronize()
lib.wf_debug_attn_cycles(buf, 1)
tiles = max(buf[6], 1)
print("w4 per tile: gaps0-31 %.1f  gaps32-63 %.1f  commit %.1f  drain %.1f  barrier %.1f cycles" % tuple(buf[i] / tiles for i in range(5)))
