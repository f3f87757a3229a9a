#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_multirank.py tests/test_gpu_rccl2.py tests/test_gpu_fullsize.py -q -x 2>&1 | grep -E "passed|failed|Error|error" > gpurun_out/ctx.txt
( P=8 python tools/dit_pair_time.py; P=8 WF_CTX_REPLICATED=1 python tools/dit_pair_time.py ) 2>&1 | grep -v amdgpu >> gpurun_out/ctx.txt
cat gpurun_out/ctx.txt
