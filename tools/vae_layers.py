"""Per-convolution time of one VAE decode + encode at the 480p size (fp32-class default): every _conv call timed with events (a sync per call:
the sum is larger than the pipelined total).  Columns: layer, entry point (333 = wf_conv3d_333 / k_conv_w4, cl = wf_conv3d_cl / k_conv, k_conv_pp), operand and output
shapes ([T,H,C/16,W,16] = the slice-major operand; channel counts include the three terms of the fp32-class mode), ms, issued TFLOP/s.  Sorted by time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from worldforge_amd import vae as wvae

DEV = "cuda:0"
rows = []
real = wvae.AutoencoderKLWan._conv


def timed(self, x, p, To, Ho, Wo, Cout, k, *a, **kw):
    f0 = self.flops_last
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = real(self, x, p, To, Ho, Wo, Cout, k, *a, **kw)
    e1.record()
    torch.cuda.synchronize()
    fam = "333" if x.dim() == 5 or (tuple(k) == (3, 3, 3) and kw.get("st", 1) == 1 and kw.get("ss", 1) == 1 and not kw.get("up2") and not kw.get("tsplit")
                                    and x.shape[-1] % 32 == 0 and Cout % 32 == 0) else "cl "   # wf_conv3d_333 (k_conv_w4) or wf_conv3d_cl (k_conv / k_conv_pp)
    rows.append((p, fam, tuple(x.shape), (To, Ho, Wo, Cout), tuple(k), e0.elapsed_time(e1), self.flops_last - f0))
    return r


if __name__ == "__main__":
    wvae.AutoencoderKLWan._conv = timed
    m = wvae.AutoencoderKLWan(DEV, precision=os.environ.get("PRECISION", "fp16x3")).init_random(seed=1)
    g = torch.Generator(device=DEV).manual_seed(0)
    z = torch.randn((1, 16, 21, 60, 104), generator=g, device=DEV)
    video = torch.rand((1, 3, 81, 480, 832), generator=g, device=DEV) * 2 - 1
    for what, fn in (("decode", lambda: m.decode(z, return_dict=False)), ("encode", lambda: m.encode(video))):
        fn(); rows.clear(); fn()
        tot = sum(r[5] for r in rows)
        print(f"== {what}: {len(rows)} convolutions, {tot:.1f} ms summed")
        agg = {}
        for r in rows:
            key = (r[0], r[1], r[2], r[3], r[4])
            a = agg.setdefault(key, [0, 0.0, 0.0]); a[0] += 1; a[1] += r[5]; a[2] += r[6]
        for key, (n, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            print(f"  {key[0]:42s} {key[1]} in={key[2]} out={key[3]} k={key[4]} x{n}: {ms:8.2f} ms {100 * ms / tot:5.1f} %  {fl / ms / 1e9:6.0f} TFLOP/s issued")
