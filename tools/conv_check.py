"""Determinism / A-B check of wf_conv3d_333 (k_conv_w4): seeded operands, each shape run `reps` times, one SHA-1 of the fp32 output per run.
Run it under two libraries (WF_LIB=...) and diff the printed lines: identical arithmetic order => identical hashes."""
import hashlib, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from worldforge_amd import _ffi, ops

BF = torch.bfloat16


def check(T, Ho, W, cin, cout, ph=1, resid=False, reps=3, seed=0):
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(seed)
    Hi = Ho if ph else Ho + 2
    x = torch.randn(T, Hi, W, cin, device=dev, generator=g).to(BF)
    x = x.view(T, Hi, W, cin // 16, 16).permute(0, 1, 3, 2, 4).contiguous()
    w = (torch.randn(cout, 27, cin, device=dev, generator=g) / math.sqrt(cin * 27)).to(BF)
    b = torch.randn(cout, device=dev, generator=g)
    r = torch.randn(T, Ho, W, cout, device=dev, generator=g) if resid else None
    zp = torch.zeros(1 << 20, dtype=BF, device=dev)
    wp = torch.empty((27, cin // 16, cout, 16), dtype=BF, device=dev)
    _ffi.call("wf_conv3d_pack333", w.data_ptr(), wp.data_ptr(), cout, cin, ops.stream())
    hs = []
    for _ in range(reps):
        out = torch.full((T, Ho, W, cout), float("nan"), device=dev)
        _ffi.call("wf_conv3d_333", x.data_ptr(), wp.data_ptr(), b.data_ptr(), r.data_ptr() if resid else None, out.data_ptr(), None, T, Hi, W,
                  cin, Ho, cout, ph, zp.data_ptr(), zp.numel() * 2, 1, cin, ops.stream())
        torch.cuda.synchronize()
        assert torch.isfinite(out).all()
        hs.append(hashlib.sha1(out.cpu().numpy().tobytes()).hexdigest()[:12])
    print(f"T={T} Ho={Ho} W={W} {cin}->{cout} ph={ph} resid={int(resid)}: {' '.join(hs)}{'' if len(set(hs)) == 1 else '  NON-DETERMINISTIC'}", flush=True)


if __name__ == "__main__":
    check(9, 480, 832, 96, 96)
    check(9, 480, 832, 96, 96, resid=True)
    check(9, 60, 832, 96, 96, ph=0)
    check(9, 240, 416, 192, 192)
    check(9, 30, 416, 192, 192, ph=0)
    check(5, 120, 208, 384, 384)
    check(5, 15, 208, 384, 384, ph=0)
    check(3, 32, 32, 96, 96)
    check(3, 60, 104, 384, 384)
    check(9, 480, 832, 288, 96)
    check(21, 60, 104, 32, 32)
