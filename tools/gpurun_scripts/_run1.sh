set -x
python -m pytest tests -m gpu -x -q > gpurun_out/r2_pytest_a.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_pytest_a.log
WF_SHARE_GPU=1 WF_COMM_BACKEND=gloo timeout 900 python bench.py --gpus 2 --layers 2 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r2_bench_2p.json 2> gpurun_out/r2_bench_2p.err; echo "rc=$?" >> gpurun_out/r2_bench_2p.err
timeout 900 python bench.py > gpurun_out/r2_bench_a.json 2> gpurun_out/r2_bench_a.err; echo "rc=$?" >> gpurun_out/r2_bench_a.err
tail -3 gpurun_out/r2_pytest_a.log; cat gpurun_out/r2_bench_2p.json; tail -2 gpurun_out/r2_bench_2p.err; cat gpurun_out/r2_bench_a.json
