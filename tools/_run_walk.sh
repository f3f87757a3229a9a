#!/bin/bash
mkdir -p gpurun_out
WF_VAE_LOG_CONV=1 python - > gpurun_out/convlog.txt 2>&1 <<'PY'
import torch
from worldforge_amd.vae import AutoencoderKLWan
dev = torch.device("cuda:0")
vae = AutoencoderKLWan(dev).init_random(seed=1)
z = torch.randn(1, 16, 21, 60, 104, device=dev)
video = torch.rand(1, 3, 81, 480, 832, device=dev) * 2 - 1
print("== decode"); vae.decode(z, return_dict=False)
print("== encode"); vae.encode(video).latent_dist.mode()
PY
grep -v amdgpu gpurun_out/convlog.txt | grep -v "k=(3, 3, 3) st=1 ss=1 up2=False tsplit=False.*layout=1"
