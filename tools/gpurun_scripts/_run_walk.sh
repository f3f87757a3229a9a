#!/bin/bash
mkdir -p gpurun_out
python bench.py --height 720 --width 1280 --steps 2 --warmup 0 --no-cpu-baseline > gpurun_out/r2_h_c3_bench.json 2> gpurun_out/c3.err
python bench.py --workload longcat --no-cpu-baseline > gpurun_out/r2_h_longcat_bench.json 2> gpurun_out/lc.err
python bench.py --workload longcat --distill --no-cpu-baseline > gpurun_out/r2_h_longcat_distill_bench.json 2> gpurun_out/lcd.err
for f in gpurun_out/r2_h_c3_bench.json gpurun_out/r2_h_longcat_bench.json gpurun_out/r2_h_longcat_distill_bench.json; do cut -c1-200 $f; done
tail -2 gpurun_out/c3.err gpurun_out/lc.err gpurun_out/lcd.err | grep -v amdgpu
