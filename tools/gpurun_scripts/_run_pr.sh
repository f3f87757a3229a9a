#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_warp.py -q -x 2>&1 | tail -25 > gpurun_out/pr.log
cat gpurun_out/pr.log
