"""Experiment: does breaking the lockstep of k_gemm_pp's workgroups hide the read-modify-write epilogue?  (all 256 workgroups
reach the epilogue together and it runs at the HBM rate while it lasts.)  The GEMM x += gate * (X W^T + b) is launched (a) as one kernel, (b) as two
column halves on two streams with the SAME tile width (control), (c) as two column halves with the 320- and the 256-feature tile: workgroups of two
durations (1 : 0.8) drift apart within a few rounds.  Needs a lab library built with -DWF_GEMM_LAB_TILE (WF_GEMM_TILE read per call):
    WF_EXTRA_HIPCC_FLAGS=-DWF_GEMM_LAB_TILE python tools/lab_lib.py gemm_labtile gemm.hip=WORK
    WF_LIB=worldforge_amd/_lib/lab/libwf_hip_gemm_labtile.so python tools/gemm_dephase.py"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from worldforge_amd import dit

DEV = "cuda:0"


def run(M, N, K, split, tiles, reps=20):
    x = torch.randn(M, K, device=DEV).bfloat16()
    w = (torch.randn(N, K, device=DEV) / math.sqrt(K)).bfloat16()
    b, g = torch.randn(N, device=DEV), torch.randn(N, device=DEV)
    out = torch.zeros(M, N, device=DEV)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def once():
        if split is None:
            os.environ["WF_GEMM_TILE"] = str(tiles[0])
            dit.gemm(x, w, b, out, 3, gate=g)
            return
        cur = torch.cuda.current_stream()
        s1.wait_stream(cur); s2.wait_stream(cur)
        for st, (c0, c1), tile in ((s1, (0, split), tiles[0]), (s2, (split, N), tiles[1])):
            with torch.cuda.stream(st):
                os.environ["WF_GEMM_TILE"] = str(tile)
                dit.gemm(x, w[c0:c1], b[c0:c1], out[:, c0:c1], 3, gate=g[c0:c1])
        cur.wait_stream(s1); cur.wait_stream(s2)

    for _ in range(3):
        once()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        once()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    return ms, 2.0 * M * N * K / ms / 1e9


if __name__ == "__main__":
    for (M, N, K) in ((32760, 5120, 5120), (32760, 5120, 13824)):
        print(f"M={M} N={N} K={K}")
        for name, split, tiles in (("one launch, 320-wide tiles", None, (320,)), ("one launch, 256-wide tiles", None, (256,)),
                                   ("two streams, 2560 + 2560 columns, 320 / 320", 2560, (320, 320)),
                                   ("two streams, 2560 + 2560 columns, 320 / 256", 2560, (320, 256)),
                                   ("two streams, 3200 + 1920 columns, 320 / 256", 3200, (320, 256)),
                                   ("two streams, 1920 + 3200 columns, 320 / 256", 1920, (320, 256))):
            ms = [run(M, N, K, split, tiles) for _ in range(2)]
            print(f"  {name}: " + "  ".join(f"{m:.3f} ms {t:.0f} TFLOP/s" for m, t in ms), flush=True)
