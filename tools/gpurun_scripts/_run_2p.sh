#!/bin/bash
mkdir -p gpurun_out
WF_SHARE_GPU=1 WF_COMM_BACKEND=gloo python bench.py --gpus 2 --layers 2 --steps 2 --warmup 1 > gpurun_out/r2_k_bench_2p.json 2> gpurun_out/2p.err; echo "rc=$?" >> gpurun_out/2p.err
wc -l gpurun_out/r2_k_bench_2p.json; cut -c1-300 gpurun_out/r2_k_bench_2p.json; tail -2 gpurun_out/2p.err
