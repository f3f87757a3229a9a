"""Micro-benchmarks of the MFMA kernels at the BASELINE config-2 shapes (L = 32760 tokens, 40 heads x 128, d = 5120).
HIP-event timing on torch's current stream.  python tools/microbench.py [attn|gemm|all] [--L N]"""
import argparse
import math
import sys
import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from worldforge_amd import dit


def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def bench_attn(L, H=40):
    dev = "cuda:0"
    Lp = (L + 63) // 64 * 64
    q = torch.randn(H, L, 128, device=dev).to(torch.bfloat16)
    k = torch.zeros(H, Lp, 128, device=dev, dtype=torch.bfloat16)
    k[:, :L] = torch.randn(H, L, 128, device=dev).to(torch.bfloat16)
    vt = torch.randn(H, Lp // 64, 128, 64, device=dev).to(torch.bfloat16)
    out = torch.empty(L, H * 128, device=dev, dtype=torch.bfloat16)
    ms = timeit(lambda: dit.attention(q, k, vt, out, L, 1 / math.sqrt(128)))
    flop = 4.0 * L * L * 128 * H
    print(f"attn L={L} H={H}: {ms:.2f} ms  {flop / ms / 1e9:.1f} TFLOP/s  ({flop / ms / 1e9 / 2500 * 100:.1f}% of 2.5 PF)")


def bench_attn_shard(L, P, H=40):
    """Per-rank self-attention of the P-way sequence-parallel plan: Lq = one token shard, keys = all L tokens."""
    from worldforge_amd.parallel import shard_plan
    dev = "cuda:0"
    plan = shard_plan(L, P)
    Lq, Lp = plan.shard_len, plan.shard_len
    q = torch.randn(H, Lq, 128, device=dev).to(torch.bfloat16)
    k = torch.randn(P, H, Lp, 128, device=dev).to(torch.bfloat16)
    vt = torch.randn(P, H, Lp // 64, 128, 64, device=dev).to(torch.bfloat16)
    out = torch.empty(Lq, H * 128, device=dev, dtype=torch.bfloat16)
    ns = dit.kv_splits(H, Lq, L)
    ms = timeit(lambda: dit.attention(q, k, vt, out, L, 1 / math.sqrt(128)))
    ms1 = timeit(lambda: dit.attention(q, k, vt, out, L, 1 / math.sqrt(128), nsplit=1))
    flop = 4.0 * Lq * L * 128 * H
    print(f"attn shard P={P} nsplit={ns} (unsplit {ms1:.2f} ms)")
    print(f"attn shard P={P} Lq={Lq} L={L}: {ms:.2f} ms  {flop / ms / 1e9:.1f} TFLOP/s  ({flop / ms / 1e9 / 2500 * 100:.1f}% of 2.5 PF)")


def bench_gemm(M, N, K, epi=0):
    dev = "cuda:0"
    x = torch.randn(M, K, device=dev).to(torch.bfloat16)
    w = (torch.randn(N, K, device=dev) / math.sqrt(K)).to(torch.bfloat16)
    b = torch.randn(N, device=dev)
    out = torch.zeros(M, N, device=dev, dtype=torch.bfloat16 if epi in (0, 1) else torch.float32)
    ms = timeit(lambda: dit.gemm(x, w, b, out, epi))
    flop = 2.0 * M * N * K
    print(f"gemm M={M} N={N} K={K} epi={epi}: {ms:.2f} ms  {flop / ms / 1e9:.1f} TFLOP/s  ({flop / ms / 1e9 / 2500 * 100:.1f}% of 2.5 PF)")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="?", default="all")
    ap.add_argument("--L", type=int, default=32760)
    a = ap.parse_args()
    if a.what in ("attn", "all"):
        for L in (4524, a.L):
            bench_attn(L)
    if a.what in ("shard",):
        for P in (2, 4, 8):
            bench_attn_shard(a.L, P)
            M = (a.L + P - 1) // P
            bench_gemm(M, 15360, 5120, 0)
            bench_gemm(M, 13824, 5120, 1)
            bench_gemm(M, 5120, 13824, 3)
    if a.what in ("gemm", "all"):
        L = a.L
        bench_gemm(L, 15360, 5120, 0)
        bench_gemm(L, 5120, 5120, 3)
        bench_gemm(L, 13824, 5120, 1)
        bench_gemm(L, 5120, 13824, 3)
        bench_gemm(4096, 4096, 4096, 0)
        bench_gemm(8192, 8192, 8192, 0)
