"""Block-sparse attention kernel alone at the refine-pass shape (32 heads, 98 560 tokens = 770 blocks, 96 selected key blocks per query
block): random selections (what a random-weight model produces) vs spatially coherent ones (a window around the query block, as the
neighbouring bricks of a video select).  python tools/bsa_bench.py"""
import sys

import torch

sys.path.insert(0, ".")
from worldforge_amd import bsa, dit  # noqa: E402

if __name__ == "__main__":
    dev = torch.device("cuda:0")
    H, nb, nsel = 32, 770, 96
    L = nb * 128
    g = torch.Generator(device=dev).manual_seed(1)
    q = torch.randn((H, L, 128), generator=g, device=dev).bfloat16()
    k = torch.randn((H, L, 128), generator=g, device=dev).bfloat16()
    vt = torch.randn((H, L // 64, 128, 64), generator=g, device=dev).bfloat16()
    out = torch.empty((L, H * 128), dtype=torch.bfloat16, device=dev)
    scale = 128 ** -0.5

    def timeit(fn, it=3):
        fn()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(it):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / it

    rnd = torch.stack([torch.stack([torch.randperm(nb, device=dev)[:nsel] for _ in range(nb)]) for _ in range(2)])
    rnd = rnd.repeat(H // 2, 1, 1)
    base = torch.arange(nb, device=dev).view(1, nb, 1) + torch.arange(nsel, device=dev).view(1, 1, nsel) - nsel // 2
    win = base.clamp(0, nb - 1)
    win = torch.where(base < 0, base + nsel, win)
    win = torch.where(base >= nb, base - nsel, win).expand(H, -1, -1).contiguous()
    flop = 4.0 * L * nsel * 128 * 128 * H
    for name, idx in (("random", rnd), ("window", win)):
        lists, counts, mx = bsa.group_lists(idx, nb)
        ms_all = timeit(lambda: bsa.sparse_attention(q, k, vt, out, idx, scale, nb))
        from worldforge_amd._ffi import call
        from worldforge_amd import ops
        ms_k = timeit(lambda: call("wf_attn_bsa_fwd", q.data_ptr(), k.data_ptr(), vt.data_ptr(), out.data_ptr(), H, L, L, L, out.stride(0), scale,
                                   lists.data_ptr(), counts.data_ptr(), mx, 128, ops.stream()))
        print(f"{name}: union {counts.float().mean().item():.1f} blocks per 2 query blocks; kernel {ms_k:.2f} ms = {flop / ms_k / 1e9:.0f} TFLOP/s of "
              f"selected work ({flop / ms_k / 1e9 * counts.float().mean().item() / nsel:.0f} walked); with list building {ms_all:.2f} ms")
    d = timeit(lambda: dit.attention(q, k, vt, out, L, scale), it=1)
    print(f"dense: {d:.1f} ms = {4.0 * L * L * 128 * H / d / 1e9:.0f} TFLOP/s")
