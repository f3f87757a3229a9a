#!/bin/bash
# round 5, f: where the row-slab VAE loses per-rank efficiency at 8 ranks (tools/vae_rank_probe.py) + one more box pair
mkdir -p gpurun_out/r5
timeout 600 python tools/vae_rank_probe.py > gpurun_out/r5/f_vae_rank_probe.txt 2>&1; cat gpurun_out/r5/f_vae_rank_probe.txt | cut -c1-1500
bash tools/gpurun_scripts/r5_box.sh f
