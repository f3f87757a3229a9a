#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_longcat.py tests/test_gpu_longcat_sampler.py tests/test_gpu_bsa.py tests/test_gpu_multirank.py tests/test_gpu_e2e.py -q -x 2>&1 | tail -12 > gpurun_out/lcps.log
( python tools/longcat_bench.py; WF_ATTN_PRESCALE=0 python tools/longcat_bench.py; python tools/longcat_bench.py ) 2>&1 | grep -v amdgpu >> gpurun_out/lcps.log
cat gpurun_out/lcps.log
