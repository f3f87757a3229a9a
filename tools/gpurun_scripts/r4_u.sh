#!/bin/bash
# round 4, u: plumbing of the N > 1 bench path at HEAD on ONE GPU (WF_SHARE_GPU=1 WF_COMM_BACKEND=gloo: all ranks on GPU 0, host-staged collectives;
# the timings mean nothing): bench.py --gpus 2 self-launched and under torch.distributed.run as the driver starts it, Wan and LongCat workloads, 2 DiT layers
#   -> gpurun_out/r4/u_*.json
mkdir -p gpurun_out/r4
export WF_SHARE_GPU=1 WF_COMM_BACKEND=gloo
timeout 900 python bench.py --gpus 2 --layers 2 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r4/u_self2.json 2> gpurun_out/r4/u_self2.err; echo "self-launched rc=$?"
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --layers 2 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r4/u_torchrun2.json 2> gpurun_out/r4/u_torchrun2.err; echo "torchrun rc=$?"
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29542 bench.py --gpus 2 --layers 2 --steps 2 --warmup 1 --no-cpu-baseline --workload longcat --distill > gpurun_out/r4/u_torchrun2_longcat.json 2> gpurun_out/r4/u_torchrun2_longcat.err; echo "torchrun longcat rc=$?"
WF_ATTN_SEGMENTED=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29543 bench.py --gpus 4 --layers 2 --steps 2 --warmup 1 --no-cpu-baseline --workload longcat --distill > gpurun_out/r4/u_torchrun4_longcat_seg.json 2> gpurun_out/r4/u_torchrun4_longcat_seg.err; echo "torchrun 4 longcat segmented rc=$?"
for f in gpurun_out/r4/u_*.json; do echo "$f: $(head -c 400 $f)"; done
tail -3 gpurun_out/r4/u_*.err | tail -20
