#!/bin/bash
mkdir -p gpurun_out/asrank8
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/asrank8/prof -- python3 bench.py --as-rank-of 8 --no-cpu-baseline > gpurun_out/asrank8/bench.json 2> gpurun_out/asrank8/err
find gpurun_out/asrank8/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/asrank8/kernel_stats.csv
find gpurun_out/asrank8/prof -type f ! -name "*stats*" -delete
head -30 gpurun_out/asrank8/kernel_stats.csv | cut -c1-150
