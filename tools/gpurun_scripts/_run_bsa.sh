#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_bsa.py -q -x 2>&1 | tail -15 > gpurun_out/bsa.log
cat gpurun_out/bsa.log
