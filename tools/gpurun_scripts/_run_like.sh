#!/bin/bash
# like-for-like with round 1's configuration (bf16-operand VAE, temporal-difference FLF gate) and the bf16-VAE line with the Farneback gate
mkdir -p gpurun_out
python bench.py --vae-precision bf16 --no-cpu-baseline > gpurun_out/r2_h_bench_bf16vae.json 2> gpurun_out/like.err
python bench.py --vae-precision bf16 --flow-backend tdiff --no-cpu-baseline > gpurun_out/r2_h_bench_r1config.json 2>> gpurun_out/like.err
python bench.py --no-cpu-baseline > gpurun_out/r2_h_bench_default_samebox.json 2>> gpurun_out/like.err
for f in gpurun_out/r2_h_bench_bf16vae.json gpurun_out/r2_h_bench_r1config.json gpurun_out/r2_h_bench_default_samebox.json; do cut -c1-120 $f; done
