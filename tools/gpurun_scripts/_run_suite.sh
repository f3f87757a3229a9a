#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|rror" | tail -3 > gpurun_out/suite.txt
python bench.py > gpurun_out/suite_bench.json 2>> gpurun_out/suite.txt
cat gpurun_out/suite.txt | grep -v amdgpu | tail -4; cut -c1-130 gpurun_out/suite_bench.json
