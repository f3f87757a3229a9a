#!/bin/bash
mkdir -p gpurun_out
WF_ATTN_KERNEL=w8 python -m pytest tests/test_gpu_vae.py tests/test_gpu_dit.py -q -x 2>&1 | tail -25 > gpurun_out/w8.txt
cat gpurun_out/w8.txt
