#!/bin/bash
# Round-end validation on one MI355X: GPU suite, contract bench, smoke, one-rank RCCL bench, rocprofv3 kernel trace of the bench.
#   gpurun --timeout 2400 -- 'bash tools/gpurun_scripts/final.sh'      -> gpurun_out/final/*
mkdir -p gpurun_out/final
rm -f gpurun_out/final/tolerances.txt
R=$GRAFT_REPO_ROOT
WF_TOL_LOG=$R/gpurun_out/final/tolerances.txt python -m pytest tests -m gpu -q -x 2>&1 | grep -E "passed|failed|rror|^FAILED|^tests/" | tail -12 > gpurun_out/final/pytest_gpu.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err   # the driver's own command (with the `also` windows of configs 3 / 4)
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/final/smoke.log
WF_FORCE_COMM=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 python bench.py --gpus 1 --no-cpu-baseline --no-also > gpurun_out/final/bench_rccl1.json 2> gpurun_out/final/bench_rccl1.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/prof -- python3 $R/bench.py --no-cpu-baseline --no-also > $R/gpurun_out/final/bench_prof.json 2> $R/gpurun_out/final/bench_prof.err
cd $R
find gpurun_out/final/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/final/kernel_stats.csv
rm -rf gpurun_out/final/prof
tail -3 gpurun_out/final/pytest_gpu.log; head -c 700 gpurun_out/final/bench.json; echo; tail -2 gpurun_out/final/smoke.log; head -c 300 gpurun_out/final/bench_rccl1.json
