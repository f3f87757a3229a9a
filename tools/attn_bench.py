"""Self-attention kernel at the C2 / C3 shapes: in-kernel scale (k_attn_w4<0>) vs pre-scaled Q (k_attn_w4<4>)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from worldforge_amd import dit
H = 40
for L in (32760, 75600):
    Lp = (L + 63) // 64 * 64
    q = torch.randn(H, L, 128, device="cuda").bfloat16()
    k = torch.zeros(H, Lp, 128, device="cuda", dtype=torch.bfloat16); k[:, :L] = torch.randn(H, L, 128, device="cuda").bfloat16()
    vt = torch.randn(H, Lp // 64, 128, 64, device="cuda").bfloat16()
    out = torch.empty(L, H * 128, device="cuda", dtype=torch.bfloat16)
    qs = (q.float() * (1.4426950408889634 / math.sqrt(128))).bfloat16()
    km, qm = dit.head_max_norm2(k, L, torch.empty(H, device="cuda")), dit.head_max_norm2(qs, L, torch.empty(H, device="cuda"))
    print("bound B per head: max", float((km * qm).sqrt().max()))
    for name, qq, sc, kk in (("in-kernel scale", q, 1 / math.sqrt(128), None), ("pre-scaled Q", qs, 0.0, None), ("pre-scaled Q + key-norm bound", qs, 0.0, km)) * 2:
        for _ in range(2): dit.attention(qq, k, vt, out, L, sc, kmax2=kk, qmax2=qm if kk is not None else None)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5): dit.attention(qq, k, vt, out, L, sc, kmax2=kk, qmax2=qm if kk is not None else None)
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 5
        print(f"L={L} {name}: {ms:.2f} ms {4.0 * L * L * 128 * H / ms / 1e9:.0f} TFLOP/s", flush=True)
