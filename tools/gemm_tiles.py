"""Per-rank DiT GEMM shapes at P = 1, 2, 4, 8 ranks, 256- vs 320-feature tiles of the ping-pong kernel (WF_GEMM_TILE=256|320 is read
once per process, so each setting runs in a child process).  python tools/gemm_tiles.py"""
import os
import subprocess
import sys

CHILD = r"""
import torch, sys
sys.path.insert(0, '.')
from worldforge_amd import dit
def t(fn, it=8, warm=3):
    for _ in range(warm): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
for P in (1, 2, 4, 8):
    M = (32760 + P - 1) // P
    for (N, K, epi) in ((15360, 5120, 0), (5120, 5120, 3), (5120, 5120, 0), (13824, 5120, 1), (5120, 13824, 3)):
        x = torch.randn(M, K, device='cuda').bfloat16(); w = (torch.randn(N, K, device='cuda') / K ** 0.5).bfloat16()
        b = torch.randn(N, device='cuda'); g = torch.randn(N, device='cuda')
        out = torch.zeros(M, N, device='cuda', dtype=torch.bfloat16 if epi < 2 else torch.float32)
        ms = t(lambda: dit.gemm(x, w, b, out, epi, gate=g if epi == 3 else None))
        print(f"P={P} M={M} N={N} K={K} epi={epi}: {ms:.3f} ms {2*M*N*K/ms/1e9:.0f} TF", flush=True)
"""

if __name__ == "__main__":
    for tile in ("256", "320", ""):
        env = dict(os.environ)
        if tile:
            env["WF_GEMM_TILE"] = tile
        print(f"--- WF_GEMM_TILE={tile or 'auto'}", flush=True)
        subprocess.run([sys.executable, "-c", CHILD], env=env, check=True)
