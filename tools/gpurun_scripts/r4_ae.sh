#!/bin/bash
# round 4, ae: rocprofv3 kernel table of one simulated rank of 8 at HEAD (bench.py --as-rank-of 8 --steps 4)   -> gpurun_out/r4/ae_kernel_stats.csv
mkdir -p gpurun_out/r4
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4/ae_prof -- python3 $R/bench.py --no-cpu-baseline --steps 4 --as-rank-of 8 > $R/gpurun_out/r4/ae_bench.json 2> $R/gpurun_out/r4/ae_bench.err
cd $R
find gpurun_out/r4/ae_prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r4/ae_kernel_stats.csv
rm -rf gpurun_out/r4/ae_prof
head -c 300 gpurun_out/r4/ae_bench.json; echo; head -30 gpurun_out/r4/ae_kernel_stats.csv | cut -c1-150
