"""Per-phase cycle accounting of the self-attention main loop (`k_attn_w4<4>`): needs a library built with -DWF_ATTN_TIMING, e.g.
    python tools/attn_lab.py build --variants timing_base
    WF_LIB=worldforge_amd/_lib/lab/libwf_hip_timing_base.so python tools/attn_timing.py
Prints shader cycles (s_memtime of wave 0 of every workgroup, averaged per KV tile) per group of 8 MFMA gaps, for even and odd tiles
(the odd ones end with the DMA drain and the workgroup barrier), for the un-tracked and the tracked softmax body.  256 cycles per
group = the MFMA pipe never waits.  The instrumentation itself costs ~10 %."""
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
qs = (q.float() * (1.4426950408889634 / math.sqrt(128))).bfloat16()
km, qm = dit.head_max_norm2(k, L, torch.empty(H, device=dev)), dit.head_max_norm2(qs, L, torch.empty(H, device=dev))
lib = _ffi.lib()
buf = (ctypes.c_ulonglong * 32)()
for body, kk in (("un-tracked", km), ("tracked", None)):
    for _ in range(2):
        dit.attention(qs, k, vt, out, L, 0.0, kmax2=kk, qmax2=qm if kk is not None else None)
    torch.cuda.synchronize()
    lib.wf_debug_attn_cycles(buf, 1)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    dit.attention(qs, k, vt, out, L, 0.0, kmax2=kk, qmax2=qm if kk is not None else None)
    b.record()
    torch.cuda.synchronize()
    lib.wf_debug_attn_cycles(buf, 1)
    pairs = max(buf[19], 1) / 2.0      # tiles counted once per workgroup (wave 0, lane 0); half of them even, half odd
    ev = [buf[i] / pairs for i in range(8)]
    od = [buf[8 + i] / pairs for i in range(8)]
    print(f"{body} body, {a.elapsed_time(b):.2f} ms per launch (instrumented)")
    print("  even tile, cycles per 8-gap group: " + " ".join(f"{v:5.0f}" for v in ev) + f"   sum {sum(ev):.0f}")
    print("  odd  tile, cycles per 8-gap group: " + " ".join(f"{v:5.0f}" for v in od) + f"   sum {sum(od):.0f}")
    print(f"  per odd tile: commit/seam {buf[16] / pairs / 2:.0f} (per tile)  drain {buf[17] / pairs:.0f}  barrier {buf[18] / pairs:.0f};"
          f"  two tiles = {(sum(ev) + sum(od) + buf[16] / pairs + buf[17] / pairs + buf[18] / pairs):.0f} cycles (ideal 4096)")
