"""One DiT CFG pair as rank P//2 of a P-rank job on one GPU (parallel.LoopbackComm), for rocprofv3 --kernel-trace: which launches do not shrink
with the token shard (work replicated on every rank)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from worldforge_amd import dit as wdit, parallel

dev = torch.device("cuda:0")
P = int(os.environ.get("P", "8"))
cfg = wdit.DiTConfig.wan_i2v_14b()
cfg.num_layers = int(os.environ.get("LAYERS", "4"))
comm = parallel.LoopbackComm(P, P // 2) if P > 1 else None
m = wdit.WanTransformer3DModel(cfg, dev, comm=comm).init_random(seed=0)
g = torch.Generator(device=dev).manual_seed(1)
x = torch.randn((36, 21, 60, 104), generator=g, device=dev).bfloat16()
text = torch.randn((512, 4096), generator=g, device=dev).bfloat16()
neg = torch.randn((512, 4096), generator=g, device=dev).bfloat16()
clip = torch.randn((257, 1280), generator=g, device=dev).bfloat16()
for _ in range(int(os.environ.get("N", "2"))):
    a, b = m.forward_tokens_pair(x, 500.0, text, neg, clip)
torch.cuda.synchronize()
print("ok", float(a.float().abs().mean()))
