"""Run the self-attention kernel a few times at the config-2 shape (for rocprofv3 --pmc passes)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from worldforge_amd import dit
L, H = int(os.environ.get("L", 32760)), 40
dev = "cuda:0"
Lp = (L + 63) // 64 * 64
q = torch.randn(H, L, 128, device=dev).to(torch.bfloat16)
k = torch.zeros(H, Lp, 128, device=dev, dtype=torch.bfloat16); k[:, :L] = torch.randn(H, L, 128, device=dev).to(torch.bfloat16)
vt = torch.randn(H, Lp // 64, 128, 64, device=dev).to(torch.bfloat16)
out = torch.empty(L, H * 128, device=dev, dtype=torch.bfloat16)
for _ in range(int(os.environ.get("N", 3))):
    dit.attention(q, k, vt, out, L, 0.0 if os.environ.get("PRESCALED") else 1 / math.sqrt(128))
torch.cuda.synchronize()
