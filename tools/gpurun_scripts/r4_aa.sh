#!/bin/bash
# round 4, aa: per-convolution time of one VAE decode + encode at 480p (tools/vae_layers.py): which layers run outside k_conv_w4 and at what rate
#   -> gpurun_out/r4/aa_vae_layers.txt
mkdir -p gpurun_out/r4
timeout 900 python tools/vae_layers.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4/aa_vae_layers.txt
cut -c1-230 gpurun_out/r4/aa_vae_layers.txt
