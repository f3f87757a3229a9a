#!/bin/bash
# per-rank step time of the N-rank job on one GPU (compute-bound ceiling of the scaling curve)
mkdir -p gpurun_out
python bench.py --no-cpu-baseline > gpurun_out/r2_i_asrank_1.json 2> gpurun_out/asrank.err
for n in 2 4 8; do python bench.py --as-rank-of $n --no-cpu-baseline > gpurun_out/r2_i_asrank_$n.json 2>> gpurun_out/asrank.err; done
for n in 1 2 4 8; do cut -c1-100 gpurun_out/r2_i_asrank_$n.json; done; tail -3 gpurun_out/asrank.err | grep -v amdgpu
