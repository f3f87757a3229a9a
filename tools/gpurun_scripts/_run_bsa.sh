#!/bin/bash
mkdir -p gpurun_out
( python tools/longcat_bench.py --bsa --frames 28 --h 88 --w 160 --iters 3; WF_BSA_TORCH_SELECT=1 python tools/longcat_bench.py --bsa --frames 28 --h 88 --w 160 --iters 3;  python tools/longcat_bench.py --bsa --frames 28 --h 88 --w 160 --iters 3 ) > gpurun_out/bsa.log 2>&1
cat gpurun_out/bsa.log
