#!/bin/bash
mkdir -p gpurun_out
( for w in 0 1 0 1; do echo "== WF_CONV_WALK=$w"; WF_CONV_WALK=$w python tools/conv_bench.py 2>&1 | grep "slice-major"; done; for w in 0 1; do echo "== WF_CONV_WALK=$w tools/vae_bench.py"; WF_CONV_WALK=$w python tools/vae_bench.py 2>&1 | grep -v amdgpu; done ) > gpurun_out/r2_conv_walk_ab.txt 2>&1
cat gpurun_out/r2_conv_walk_ab.txt
