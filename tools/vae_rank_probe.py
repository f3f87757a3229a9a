"""Per-rank VAE time of a P-rank job on one GPU (parallel.LoopbackComm: collectives served from local data) next to the single-GPU time, with
the per-convolution event timings of the sharded decode: where the row-slab VAE loses scaling efficiency."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
from benchlib.measure import synthetic_inputs
from worldforge_amd import parallel, vae as wvae
from worldforge_amd.vae import AutoencoderKLWan

dev = torch.device("cuda:0")
P = int(os.environ.get("P", "8"))
z = torch.randn(1, 16, 21, 60, 104, device=dev)
video = torch.rand(1, 3, 81, 480, 832, device=dev) * 2 - 1


def timed(fn, n=2):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n


v1 = AutoencoderKLWan(dev).init_random(seed=1)
d1, e1 = timed(lambda: v1.decode(z, return_dict=False)[0]), timed(lambda: v1.encode(video).latent_dist.mode())
vp = AutoencoderKLWan(dev, comm=parallel.LoopbackComm(P, P // 2))
vp.w = v1.w
dp, ep = timed(lambda: vp.decode(z, return_dict=False)[0]), timed(lambda: vp.encode(video).latent_dist.mode())
print(f"single GPU: decode {d1:.1f} ms, encode {e1:.1f} ms; rank {P // 2} of {P}: decode {dp:.1f} ms (x{d1 / dp:.2f}), encode {ep:.1f} ms (x{e1 / ep:.2f})")
# what an IRR injection runs: decode -> blend -> encode; on the sharded VAE the decoded row slab is never gathered.  With SURVEY 8d's mask
# (hole growing to 35 % of the width) only the columns the blend can see are decoded (vae.decode(columns=...)); CROP off = everything
_, ref, mask, _, _, _ = synthetic_inputs(81, 480, 832, dev)
for crop in (False, True):
    v1.crop_to_mask = vp.crop_to_mask = crop
    r1, rp = timed(lambda: v1.decode_blend_encode(z, ref, mask)), timed(lambda: vp.decode_blend_encode(z, ref, mask))
    print(f"injection round trip (decode -> blend -> encode), needed columns only = {crop}: single GPU {r1:.1f} ms; rank {P // 2} of {P}: {rp:.1f} ms (x{r1 / rp:.2f})", flush=True)
v1.crop_to_mask = vp.crop_to_mask = True

# per-call timing of the conv entry points in one sharded decode
ev = []
orig = wvae.call


def tcall(name, *a):
    if name.startswith("wf_conv3d") or name.startswith("wf_rms_silu") or name.startswith("wf_gemm") or name.startswith("wf_split"):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); orig(name, *a); e.record()
        ev.append((name, s, e, a))
    else:
        orig(name, *a)


for label, model in (("single", v1), (f"rank of {P}", vp)):
    ev.clear()
    wvae.call = tcall
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record(); model.decode(z, return_dict=False); t1.record()
    wvae.call = orig
    torch.cuda.synchronize()
    tot = {}
    for name, s, e, a in ev:
        tot[name] = tot.get(name, 0.0) + s.elapsed_time(e)
    inside = sum(tot.values())
    print(f"{label}: decode {t0.elapsed_time(t1):.1f} ms, of which timed entry points {inside:.1f} ms: " + ", ".join(f"{k} {v:.1f}" for k, v in sorted(tot.items(), key=lambda kv: -kv[1])))
