"""One DiT GEMM shape a few times with random or zero operands (for rocprofv3 --pmc clock / MFMA-busy passes).
env: SHAPE=qkv|o|ffn_up|ffn_down, ZERO=0|1, N=3."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from worldforge_amd import dit
M = 32760
N_, K, epi = {"qkv": (15360, 5120, 0), "o": (5120, 5120, 3), "ffn_up": (13824, 5120, 1), "ffn_down": (5120, 13824, 3)}[os.environ.get("SHAPE", "qkv")]
zero = int(os.environ.get("ZERO", 0))
x = (torch.zeros if zero else torch.randn)(M, K, device="cuda").bfloat16()
w = ((torch.zeros if zero else torch.randn)(N_, K, device="cuda") / K ** 0.5).bfloat16()
b, g = torch.randn(N_, device="cuda"), torch.randn(N_, device="cuda")
out = torch.zeros(M, N_, device="cuda", dtype=torch.bfloat16 if epi < 2 else torch.float32)
for _ in range(int(os.environ.get("N", 3))):
    dit.gemm(x, w, b, out, epi, gate=g if epi == 3 else None)
torch.cuda.synchronize()
