#!/bin/bash
mkdir -p gpurun_out
python tools/vae_rank_probe.py 2>&1 | grep -v amdgpu > gpurun_out/vaerank.txt
cat gpurun_out/vaerank.txt
