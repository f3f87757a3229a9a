#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_dit.py -q -x 2>&1 | tail -3 > gpurun_out/fuse_ab.txt
( for r in 1 2; do for p in 1 8; do P=$p python tools/dit_pair_time.py; done; done ) 2>&1 | grep -v amdgpu >> gpurun_out/fuse_ab.txt
cat gpurun_out/fuse_ab.txt
