"""Per-segment cycle accounting of k_conv_w4 (needs a -DWF_CONV_TIMING [-DWF_CONV_ABLATE] build: WF_LIB=worldforge_amd/_lib/lab/libwf_hip_convtiming.so;
with the ablate build WF_CONV_DEBUG=<bits> picks the variant: 1 no in-loop LDS-DMA, 2 no in-loop weight loads, 4 no LDS fragment reads).
LAYOUT=0|1 (pixel- / slice-major operand), X3=0|1 (three-term fp32-class operand)."""
import ctypes, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from worldforge_amd import _ffi, ops
lib = _ffi.lib()
buf = (ctypes.c_ulonglong * 8)()
sl = (ctypes.c_ulonglong * 32)()
LAYOUT = int(os.environ.get("LAYOUT", 1))
X3 = int(os.environ.get("X3", 0))
BF = torch.bfloat16
for (T, H, W, C) in ((81, 480, 832, 96), (81, 240, 416, 192), (41, 120, 208, 384)):
    K = 3 * C if X3 else C
    Cs = (2 * C if LAYOUT else 3 * C) if X3 else C
    x = torch.randn(T, H, W, Cs, device="cuda:0").to(BF)
    if LAYOUT:
        x = x.view(T, H, W, Cs // 16, 16).permute(0, 1, 3, 2, 4).contiguous()
    w = (torch.randn(C, 27, K, device="cuda:0") / math.sqrt(K * 27)).to(BF)
    out = torch.empty(T, H, W, C, device="cuda:0")
    bias = torch.randn(C, device="cuda:0")  # bias + fp32 out = the epilogue every ResidualBlock conv of the VAE runs
    zp = torch.zeros(1 << 20, dtype=BF, device="cuda:0")
    wp = torch.empty((27, K // 16, C, 16), dtype=BF, device="cuda:0")
    _ffi.call("wf_conv3d_pack333", w.data_ptr(), wp.data_ptr(), C, K, ops.stream())

    def run():
        _ffi.call("wf_conv3d_333", x.data_ptr(), wp.data_ptr(), bias.data_ptr(), None, out.data_ptr(), None, T, H, W, K, H, C, 1, zp.data_ptr(),
                  zp.numel() * 2, LAYOUT, Cs, ops.stream())
    run(); torch.cuda.synchronize(); lib.wf_debug_conv_cycles(buf, 1); lib.wf_debug_conv_slices(sl)
    run(); torch.cuda.synchronize(); lib.wf_debug_conv_cycles(buf, 1); lib.wf_debug_conv_slices(sl)
    nwg, nt = max(buf[5], 1), max(buf[6], 1)
    print(f"C={C} layout={LAYOUT} x3={X3}: cold start per workgroup {buf[0]/nwg:.0f};  per tile: main {buf[1]/nt:.0f} ({buf[1]/max(buf[4],1):.0f} per slice, "
          f"of which wait+barrier {buf[3]/max(buf[4],1):.0f})  epilogue {buf[2]/nt:.0f} cycles; tiles per workgroup {nt/nwg:.1f}")
    print("   per-slice cycles:", " ".join(f"{sl[i] / max(buf[6], 1):.0f}" for i in range(min(K // 16, 32))), flush=True)
