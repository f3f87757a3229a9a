"""Per-segment cycle accounting of k_conv_w4 (needs WF_EXTRA_HIPCC_FLAGS=-DWF_CONV_TIMING)."""
import ctypes, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from worldforge_amd import _ffi, ops
lib = _ffi.lib()
buf = (ctypes.c_ulonglong * 8)()
sl = (ctypes.c_ulonglong * 32)()
LAYOUT = int(os.environ.get("LAYOUT", 1))
for (T, H, W, C) in ((81, 480, 832, 96), (41, 120, 208, 384)):
    x = torch.randn(T, H, W, C, device="cuda:0").to(torch.bfloat16)
    if LAYOUT:
        x = x.view(T, H, W, C // 16, 16).permute(0, 1, 3, 2, 4).contiguous()
    w = (torch.randn(C, 27, C, device="cuda:0") / math.sqrt(C * 27)).to(torch.bfloat16)
    out = torch.empty(T, H, W, C, device="cuda:0")
    zp = torch.zeros(1024, dtype=torch.bfloat16, device="cuda:0")
    wp = torch.empty((27, C // 16, C, 16), dtype=torch.bfloat16, device="cuda:0")
    _ffi.call("wf_conv3d_pack333", w.data_ptr(), wp.data_ptr(), C, C, ops.stream())
    def run():
        _ffi.call("wf_conv3d_333", x.data_ptr(), wp.data_ptr(), None, None, out.data_ptr(), None, T, H, W, C, H, C, 1, zp.data_ptr(),
                  LAYOUT, C, ops.stream())
    run(); torch.cuda.synchronize(); lib.wf_debug_conv_cycles(buf, 1); lib.wf_debug_conv_slices(sl)
    run(); torch.cuda.synchronize(); lib.wf_debug_conv_cycles(buf, 1); lib.wf_debug_conv_slices(sl)
    nwg, nt = max(buf[5], 1), max(buf[6], 1)
    print(f"C={C}: cold start per workgroup {buf[0]/nwg:.0f};  per tile: main {buf[1]/nt:.0f} ({buf[1]/max(buf[4],1):.0f} per slice, of which wait+barrier {buf[3]/max(buf[4],1):.0f})  epilogue {buf[2]/nt:.0f} cycles")
    print("   per-slice cycles:", " ".join(f"{sl[i] / max(buf[6], 1):.0f}" for i in range(C // 16)))
