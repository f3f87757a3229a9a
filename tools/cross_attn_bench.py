"""The DiT's two cross-attentions per layer (257 CLIP + 512 text keys, model.py:202-229): two launches of k_attn_w4<1> (the second
accumulating into O) against the fused wf_attn_cross2_fwd (k_attn_w4<5>).  L = query tokens (env L, default 32760 and 4096), 40 heads."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from worldforge_amd import dit

dev, H, bf = "cuda:0", 40, torch.bfloat16
scale = 1 / math.sqrt(128)
for L in [int(v) for v in os.environ.get("L", "32760,4096").split(",")]:
    q = torch.randn(H, L, 128, device=dev).to(bf)
    kc = torch.zeros(H, 320 + 512, 128, device=dev, dtype=bf)
    kc[:, :257] = torch.randn(H, 257, 128, device=dev).to(bf)
    kc[:, 320:] = torch.randn(H, 512, 128, device=dev).to(bf)
    vtc = torch.randn(H, 13, 128, 64, device=dev).to(bf)
    ki, kt = kc[:, :320].contiguous(), kc[:, 320:].contiguous()
    vi, vt = vtc[:, :5].contiguous(), vtc[:, 5:].contiguous()
    o1, o2 = torch.empty(L, H * 128, device=dev, dtype=bf), torch.empty(L, H * 128, device=dev, dtype=bf)

    def two():
        dit.attention(q, ki, vi, o1, 257, scale)
        dit.attention(q, kt, vt, o1, 512, scale, accumulate=True)

    def one():
        dit.cross_attention2(q, kc, vtc, o2, 320, 257, 512, scale)

    res = {}
    for name, fn in (("two launches", two), ("fused", one)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1) / 20
    flop = 4.0 * L * 769 * 128 * H
    print(f"L={L}: two launches {res['two launches'] * 1e3:.0f} us ({flop / res['two launches'] / 1e9:.0f} TFLOP/s), fused {res['fused'] * 1e3:.0f} us "
          f"({flop / res['fused'] / 1e9:.0f} TFLOP/s), equal: {torch.equal(o1, o2)}")
