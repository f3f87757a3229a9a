"""Same-box A/B of two builds of the library on the DiT's GEMM shapes (all epilogues, 1 and 8 ranks), with output checksums:
    python tools/gemm_epilogue_ab.py <libA.so> <libB.so>"""
import os, subprocess, sys
CHILD = r"""
import torch, sys
sys.path.insert(0, '.')
from worldforge_amd import dit
def t(fn, it=10, warm=3):
    for _ in range(warm): fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(it): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / it
torch.manual_seed(0)
for P in (1, 8):
    M = (32760 + P - 1) // P
    for (N, K, epi) in ((15360, 5120, 0), (5120, 5120, 3), (5120, 5120, 0), (13824, 5120, 1), (5120, 13824, 3), (5120, 5120, 2), (5120, 5120, 4)):
        x = torch.randn(M, K, device='cuda').bfloat16(); w = (torch.randn(N, K, device='cuda') / K ** 0.5).bfloat16()
        b = torch.randn(N, device='cuda'); g = torch.randn(N, device='cuda')
        out = torch.zeros(M, N, device='cuda', dtype=torch.bfloat16 if epi < 2 else torch.float32)
        ms = t(lambda: dit.gemm(x, w, b, out, epi, gate=g if epi == 3 else None))
        out.fill_(0.5); dit.gemm(x, w, b, out, epi, gate=g if epi == 3 else None); torch.cuda.synchronize()
        print(f"P={P} M={M} N={N} K={K} epi={epi}: {ms:.3f} ms {2*M*N*K/ms/1e9:.0f} TF  checksum {float(out.float().double().sum()):.9e} {float(out.float().double().abs().sum()):.9e}", flush=True)
"""
for rnd in range(2):
    for lib in sys.argv[1:]:
        print(f"--- {lib} (round {rnd})", flush=True)
        subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, WF_LIB=lib), check=True)
