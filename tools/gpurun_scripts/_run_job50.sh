#!/bin/bash
mkdir -p gpurun_out
python bench.py --steps 50 --warmup 0 --no-cpu-baseline > gpurun_out/r2_h_job50_bench.json 2> gpurun_out/job50.err
cut -c1-200 gpurun_out/r2_h_job50_bench.json; tail -2 gpurun_out/job50.err | grep -v amdgpu
