#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_vae.py tests/test_gpu_multirank.py tests/test_gpu_rccl2.py -q -x 2>&1 | grep -E "passed|failed|Error" > gpurun_out/vaerank.txt
python tools/vae_rank_probe.py 2>&1 | grep -v amdgpu >> gpurun_out/vaerank.txt
cat gpurun_out/vaerank.txt
