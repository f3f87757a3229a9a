#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_flow.py tests/test_gpu_longcat_sampler.py tests/test_gpu_e2e.py -q -x 2>&1 | tail -15 > gpurun_out/lcflf.log
cat gpurun_out/lcflf.log
