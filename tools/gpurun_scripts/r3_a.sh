set -x
mkdir -p gpurun_out/r3
nproc; free -g | head -2
python -m pytest tests/test_gpu_timed_kernel_parity.py -m gpu -q -x -s --durations=10 > gpurun_out/r3/a_parity.log 2>&1; echo "rc=$?" >> gpurun_out/r3/a_parity.log
tail -30 gpurun_out/r3/a_parity.log
python bench.py > gpurun_out/r3/a_bench.json 2> gpurun_out/r3/a_bench.err; echo "bench rc=$?"
cat gpurun_out/r3/a_bench.json; tail -5 gpurun_out/r3/a_bench.err
