#!/bin/bash
mkdir -p gpurun_out
( for r in 0 1 0 1; do echo "== WF_CONV_REMAP=$r"; WF_CONV_REMAP=$r python tools/vae_bench.py; done ) 2>&1 | grep -v amdgpu.ids > gpurun_out/vaeb.txt
cat gpurun_out/vaeb.txt
