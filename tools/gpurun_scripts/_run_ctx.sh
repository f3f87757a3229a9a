#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_multirank.py tests/test_gpu_longcat.py tests/test_gpu_bsa.py -q -x 2>&1 | grep -E "passed|failed|Error|error" > gpurun_out/ctx.txt
cat gpurun_out/ctx.txt
