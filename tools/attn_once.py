"""Run the DiT self-attention kernel a few times at the config-2 shape (for rocprofv3 --pmc passes).
env: L (32760), N launches (3), MODE = ps (default: pre-scaled Q + norm bounds = k_attn_w4<4>, un-tracked body, what the DiT runs) |
     ps_tracked (k_attn_w4<4>, tracked body) | scale (k_attn_w4<0>, softmax scale inside the kernel)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from worldforge_amd import dit
L, H = int(os.environ.get("L", 32760)), 40
mode = os.environ.get("MODE", "ps")
dev = "cuda:0"
Lp = (L + 63) // 64 * 64
scale = 1 / math.sqrt(128)
q = torch.randn(H, L, 128, device=dev)
if mode != "scale":
    q = q * (scale * 1.4426950408889634)      # what wf_rmsnorm_heads(out_scale) hands over
q = q.to(torch.bfloat16)
k = torch.zeros(H, Lp, 128, device=dev, dtype=torch.bfloat16); k[:, :L] = torch.randn(H, L, 128, device=dev).to(torch.bfloat16)
vt = torch.randn(H, Lp // 64, 128, 64, device=dev).to(torch.bfloat16)
out = torch.empty(L, H * 128, device=dev, dtype=torch.bfloat16)
km = qm = None
if mode == "ps":
    km, qm = dit.head_max_norm2(k, L, torch.empty(H, device=dev)), dit.head_max_norm2(q, L, torch.empty(H, device=dev))
for _ in range(int(os.environ.get("N", 3))):
    dit.attention(q, k, vt, out, L, scale if mode == "scale" else 0.0, kmax2=km, qmax2=qm)
torch.cuda.synchronize()
