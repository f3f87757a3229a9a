"""Run wf_conv3d_333 a few times on one VAE shape (for rocprofv3 --pmc passes).  env: C (96), X3 (1), N (2), F16 (0; 1 = the fp16-operand
instantiation the shipped fp16x3 VAE runs: wf_conv3d_333_f16 with the accumulator scale of the power-of-two weight scaling)."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from worldforge_amd import _ffi, ops
BF = torch.bfloat16
F16 = int(os.environ.get("F16", 0))
if F16:
    BF = torch.float16   # same 2-byte operand layout, fp16 values
C, X3, N = int(os.environ.get("C", 96)), int(os.environ.get("X3", 1)), int(os.environ.get("N", 2))
T, H, W = {96: (81, 480, 832), 192: (81, 240, 416), 384: (41, 120, 208)}[C]
K, Cs = (3 * C, 2 * C) if X3 else (C, C)
x = torch.randn(T, H, W, Cs, device="cuda:0").to(BF).view(T, H, W, Cs // 16, 16).permute(0, 1, 3, 2, 4).contiguous()
w = (torch.randn(C, 27, K, device="cuda:0") / math.sqrt(K * 27)).to(BF)
b = torch.randn(C, device="cuda:0")
out = torch.empty(T, H, W, C, device="cuda:0")
zp = torch.zeros(1 << 20, dtype=BF, device="cuda:0")
wp = torch.empty((27, K // 16, C, 16), dtype=BF, device="cuda:0")
_ffi.call("wf_conv3d_pack333", w.data_ptr(), wp.data_ptr(), C, K, ops.stream())
for _ in range(N):
    if F16:
        _ffi.call("wf_conv3d_333_f16", x.data_ptr(), wp.data_ptr(), b.data_ptr(), None, out.data_ptr(), None, T, H, W, K, H, C, 1, zp.data_ptr(), zp.numel() * 2, 1,
                  Cs, 2.0 ** -10, ops.stream())
    else:
        _ffi.call("wf_conv3d_333", x.data_ptr(), wp.data_ptr(), b.data_ptr(), None, out.data_ptr(), None, T, H, W, K, H, C, 1, zp.data_ptr(), zp.numel() * 2, 1, Cs,
                  ops.stream())
torch.cuda.synchronize()
