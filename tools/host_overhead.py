"""Host-side enqueue time of one DiT forward (Python + ctypes + launches) vs its GPU time: at 8 ranks the GPU time per forward shrinks
~8x while the enqueue time stays, so it must stay well below it.  python tools/host_overhead.py"""
import sys
import time

import torch

sys.path.insert(0, ".")
from worldforge_amd import dit  # noqa: E402

if __name__ == "__main__":
    dev = torch.device("cuda:0")
    cfg = dit.DiTConfig.wan_i2v_14b()
    m = dit.WanTransformer3DModel(cfg, dev).init_random(0)
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn((36, 21, 60, 104), generator=g, device=dev).bfloat16()
    text = torch.randn((512, 4096), generator=g, device=dev).bfloat16()
    img = torch.randn((257, 1280), generator=g, device=dev).bfloat16()
    m.forward_tokens(x, 500.0, text, img)
    torch.cuda.synchronize()
    for _ in range(2):
        t0 = time.perf_counter()
        m.forward_tokens(x, 500.0, text, img)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"enqueue {1e3 * (t1 - t0):.1f} ms, forward complete {1e3 * (t2 - t0):.1f} ms", flush=True)
    # the same with a short sequence: GPU time ~ what one of 8 ranks has per forward
    xs = torch.randn((36, 21, 60, 104 // 8 * 1), generator=g, device=dev).bfloat16() if False else torch.randn((36, 3, 60, 104), generator=g, device=dev).bfloat16()
    m.forward_tokens(xs, 500.0, text, img)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m.forward_tokens(xs, 500.0, text, img)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"short sequence (4680 tokens): enqueue {1e3 * (t1 - t0):.1f} ms, complete {1e3 * (t2 - t0):.1f} ms")
