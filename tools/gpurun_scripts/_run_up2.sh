#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_vae.py tests/test_gpu_multirank.py tests/test_gpu_rccl2.py -q -x 2>&1 | tail -12 > gpurun_out/up2.log
( python tools/vae_bench.py; WF_VAE_UP2_PHASES=0 python tools/vae_bench.py ) 2>&1 | grep -v amdgpu >> gpurun_out/up2.log
cat gpurun_out/up2.log
