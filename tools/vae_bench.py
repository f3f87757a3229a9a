"""VAE decode + encode at the C2 size (81 x 480 x 832), default fp32-class operands (fp16x3), the one-term fp16 mode and the bf16 mode: ms per call and MFMA TFLOP/s issued."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
from benchlib.measure import synthetic_inputs
from worldforge_amd.vae import AutoencoderKLWan

dev = torch.device("cuda:0")
for prec in (sys.argv[1:] or ["fp32", "fp16", "bf16"]):
    vae = AutoencoderKLWan(dev, precision=prec).init_random(seed=1)
    z = torch.randn(1, 16, 21, 60, 104, device=dev)
    video = torch.rand(1, 3, 81, 480, 832, device=dev) * 2 - 1
    mask = synthetic_inputs(81, 480, 832, dev)[2]                  # the hole grows to 35 % of the width
    cols = vae.needed_columns(mask)
    print(f"{prec}: needed latent columns {cols} of 104 -> decoded after the latent-resolution stage: {vae._crop_range(cols, 104)}", flush=True)
    for name, fn in (("decode", lambda: vae.decode(z, return_dict=False)[0]), ("decode (needed columns only)", lambda: vae.decode(z, return_dict=False, columns=cols)[0]),
                     ("encode", lambda: vae.encode(video).latent_dist.mode())):
        fn(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(2): fn()
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 2
        print(f"{prec} {name}: {ms:.1f} ms, {vae.flops_last / ms / 1e9:.0f} TFLOP/s issued", flush=True)
    del vae
