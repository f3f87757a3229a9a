"""Per-segment cycle accounting of k_gemm_w4 (needs WF_EXTRA_HIPCC_FLAGS=-DWF_GEMM_TIMING and WF_GEMM_KERNEL=w4)."""
import ctypes, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from worldforge_amd import _ffi, dit
M, N, K = 32760, 15360, 5120
x = torch.randn(M, K, device="cuda:0").to(torch.bfloat16)
w = (torch.randn(N, K, device="cuda:0") / math.sqrt(K)).to(torch.bfloat16)
out = torch.empty(M, N, device="cuda:0", dtype=torch.bfloat16)
lib = _ffi.lib()
buf = (ctypes.c_ulonglong * 16)()
dit.gemm(x, w, None, out, 0); torch.cuda.synchronize(); lib.wf_debug_gemm_cycles(buf, 1)
dit.gemm(x, w, None, out, 0); torch.cuda.synchronize(); lib.wf_debug_gemm_cycles(buf, 1)
n = max(buf[4], 1)
print("per K tile: ks0-2 %.1f  drain %.1f  barrier %.1f  ks3 %.1f cycles" % tuple(buf[i] / n for i in range(4)))
print("per output tile: prologue %.0f  epilogue %.0f cycles; K tiles per output tile %.1f" % (buf[5] / max(buf[7], 1), buf[6] / max(buf[7], 1), buf[4] / max(buf[7], 1)))
