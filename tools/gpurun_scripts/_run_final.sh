#!/bin/bash
# round-end validation on one MI355X: GPU suite, contract bench, smoke, one-rank RCCL bench, rocprofv3 kernel trace of the bench
mkdir -p gpurun_out/final
cd /root/repo
python -m pytest tests -m gpu -q -x 2>&1 | tail -8 > gpurun_out/final/pytest_gpu.log
python bench.py > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/final/smoke.log
WF_FORCE_COMM=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 python bench.py --gpus 1 --no-cpu-baseline > gpurun_out/final/bench_rccl1.json 2> gpurun_out/final/bench_rccl1.err
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/prof -- python3 bench.py --no-cpu-baseline > gpurun_out/final/bench_prof.json 2> gpurun_out/final/bench_prof.err
find gpurun_out/final/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/final/kernel_stats.csv
find gpurun_out/final/prof -type f ! -name "*stats*" -delete
tail -3 gpurun_out/final/pytest_gpu.log; cat gpurun_out/final/bench.json | head -c 600; tail -2 gpurun_out/final/smoke.log; head -c 400 gpurun_out/final/bench_rccl1.json; tail -3 gpurun_out/final/bench_rccl1.err
