"""Micro-benchmark of wf_conv3d_333 (k_conv_w4) on the VAE's FLOP-heavy shapes: pixel-major vs slice-major operand, bf16 and fp32-class
(three-term) operands.  TFLOP/s are MFMA flops actually issued (the three-term mode issues 3x the layer's algorithmic flops)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from worldforge_amd import _ffi, ops

BF = torch.bfloat16


def bench(T, H, W, cin, cout, layout=1, x3=False, iters=3, resid=False):
    dev = "cuda:0"
    K = 3 * cin if x3 else cin          # contraction channels
    Cs = 2 * cin if x3 else cin         # stored channels ([hi | lo] for the three-term operand)
    if layout == 0 and x3:
        Cs = K                          # pixel-major three-term operand is stored [hi | lo | hi]
    x = torch.randn(T, H, W, Cs, device=dev).to(BF)
    if layout == 1:
        x = x.view(T, H, W, Cs // 16, 16).permute(0, 1, 3, 2, 4).contiguous()
    w = (torch.randn(cout, 27, K, device=dev) / math.sqrt(K * 27)).to(BF)
    b = torch.randn(cout, device=dev)
    r = torch.randn(T, H, W, cout, device=dev) if resid else None
    out = torch.empty(T, H, W, cout, device=dev)
    zp = torch.zeros(1 << 20, dtype=BF, device=dev)
    wp = torch.empty((27, K // 16, cout, 16), dtype=BF, device=dev)
    _ffi.call("wf_conv3d_pack333", w.data_ptr(), wp.data_ptr(), cout, K, ops.stream())

    def run():
        _ffi.call("wf_conv3d_333", x.data_ptr(), wp.data_ptr(), b.data_ptr(), r.data_ptr() if resid else None, out.data_ptr(), None, T, H, W,
                  K, H, cout, 1, zp.data_ptr(), zp.numel() * 2, layout, Cs, ops.stream())
    run(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): run()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    fl = 2.0 * T * H * W * cout * K * 27
    print(f"conv333 T={T} {H}x{W} {cin}->{cout} layout={'slice-major' if layout else 'pixel-major'} {'x3' if x3 else 'bf16'}"
          f"{' +resid' if resid else ''}: {ms:.2f} ms {fl / ms / 1e9:.0f} TFLOP/s issued", flush=True)
    return fl / ms / 1e9


if __name__ == "__main__":
    for x3 in (False, True):
        for layout in (0, 1):
            bench(81, 480, 832, 96, 96, layout, x3)
            bench(81, 240, 416, 192, 192, layout, x3)
            bench(41, 120, 208, 384, 384, layout, x3)
    bench(81, 480, 832, 96, 96, 1, False, resid=True)
