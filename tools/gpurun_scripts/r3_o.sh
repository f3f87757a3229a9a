set -x
mkdir -p gpurun_out/r3
WF_SHARE_GPU=1 WF_COMM_BACKEND=gloo timeout 900 python bench.py --gpus 2 --layers 2 --steps 3 --warmup 1 > gpurun_out/r3/o_bench_2p_gloo.json 2> gpurun_out/r3/o_bench_2p_gloo.err; echo "rc=$?"
cat gpurun_out/r3/o_bench_2p_gloo.json | head -c 1500; tail -5 gpurun_out/r3/o_bench_2p_gloo.err
WF_FORCE_COMM=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 python bench.py --gpus 1 --no-cpu-baseline > gpurun_out/r3/o_bench_rccl1.json 2> gpurun_out/r3/o_bench_rccl1.err; echo "rc=$?"
head -c 600 gpurun_out/r3/o_bench_rccl1.json; tail -3 gpurun_out/r3/o_bench_rccl1.err
