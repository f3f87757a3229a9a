#!/bin/bash
mkdir -p gpurun_out
python bench.py --workload longcat --no-cpu-baseline > gpurun_out/r2_i_longcat_bench.json 2> gpurun_out/lc.err
python bench.py --workload longcat --distill --no-cpu-baseline > gpurun_out/r2_i_longcat_distill_bench.json 2> gpurun_out/lcd.err
cut -c1-160 gpurun_out/r2_i_longcat_bench.json; cut -c1-160 gpurun_out/r2_i_longcat_distill_bench.json
