"""Time of one CFG pair of DiT forwards (full depth) on one GPU, as a single GPU or as rank P//2 of P (parallel.LoopbackComm)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from worldforge_amd import dit as wdit, parallel
dev = torch.device("cuda:0")
P = int(os.environ.get("P", "1"))
cfg = wdit.DiTConfig.wan_i2v_14b()
comm = parallel.LoopbackComm(P, P // 2) if P > 1 else None
m = wdit.WanTransformer3DModel(cfg, dev, comm=comm).init_random(seed=0)
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn((36, 21, 60, 104), generator=g, device=dev).bfloat16()
text, neg = (torch.randn((512, 4096), generator=g, device=dev).bfloat16() for _ in range(2))
clip = torch.randn((257, 1280), generator=g, device=dev).bfloat16()
m.forward_tokens_pair(x, 500.0, text, neg, clip); torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(3): m.forward_tokens_pair(x, 500.0, text, neg, clip)
e.record(); torch.cuda.synchronize()
print(f"P={P}: {s.elapsed_time(e) / 3:.1f} ms per CFG pair", flush=True)
