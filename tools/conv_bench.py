"""Micro-benchmark of the implicit-GEMM conv kernels on the VAE's FLOP-heavy shapes."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from worldforge_amd import _ffi, ops

def bench(T, H, W, cin, cout, k=(3, 3, 3), iters=3):
    dev = "cuda:0"
    x = torch.randn(T, H, W, cin, device=dev).to(torch.bfloat16)
    w = (torch.randn(cout, k[0] * k[1] * k[2], cin, device=dev) / math.sqrt(cin * 27)).to(torch.bfloat16)
    b = torch.randn(cout, device=dev)
    out = torch.empty(T, H, W, cout, device=dev)
    zp = torch.zeros(64, dtype=torch.bfloat16, device=dev)
    wp = torch.empty((27, cin // 16, cout, 16), dtype=torch.bfloat16, device=dev)
    _ffi.call("wf_conv3d_pack333", w.data_ptr(), wp.data_ptr(), cout, cin, ops.stream())
    def run():
        if os.environ.get("WF_CONV_NO_W4"):
            _ffi.call("wf_conv3d_cl", x.data_ptr(), w.data_ptr(), b.data_ptr(), None, out.data_ptr(), None, T, H, W, cin, T, H, W, cout,
                      k[0], k[1], k[2], 1, 1, k[0] - 1, k[1] // 2, k[2] // 2, 0, 0, zp.data_ptr(), ops.stream())
        else:
            _ffi.call("wf_conv3d_333", x.data_ptr(), wp.data_ptr(), b.data_ptr(), None, out.data_ptr(), None, T, H, W, cin, H, cout, 1,
                      zp.data_ptr(), ops.stream())
    run(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): run()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    fl = 2.0 * T * H * W * cout * cin * k[0] * k[1] * k[2]
    print(f"conv T={T} {H}x{W} {cin}->{cout} k={k}: {ms:.2f} ms {fl / ms / 1e9:.0f} TFLOP/s")

if __name__ == "__main__":
    bench(81, 480, 832, 96, 96)
    bench(81, 240, 416, 192, 192)
    bench(41, 120, 208, 384, 384)
