#!/bin/bash
# round 5, h: robustness of the new exchange paths (streams / events / in-place collectives): the exchange, multi-rank, RCCL and LongCat test
# files three times over, then the whole GPU suite with the test files in reverse order; comm_probe plumbing on two ranks sharing the GPU
#   -> gpurun_out/r5/h_*
mkdir -p gpurun_out/r5
for i in 1 2 3; do
  timeout 900 python -m pytest tests/test_gpu_exchange.py tests/test_gpu_multirank.py tests/test_gpu_rccl2.py tests/test_gpu_longcat.py -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|error" | tail -1
done > gpurun_out/r5/h_repeats.log 2>&1
cat gpurun_out/r5/h_repeats.log
timeout 1500 python -m pytest $(ls tests/test_gpu_*.py | sort -r) -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|error|^FAILED" | tail -4 > gpurun_out/r5/h_reverse.log; cat gpurun_out/r5/h_reverse.log
if [ "$1" != "tests-only" ]; then
WF_SHARE_GPU=1 WF_COMM_BACKEND=gloo timeout 900 python tools/comm_probe.py --gpus 2 --iters 2 --layers 2 > gpurun_out/r5/h_comm_probe_gloo2.json 2> gpurun_out/r5/h_comm_probe_gloo2.err; echo "comm_probe rc=$?"
head -c 1500 gpurun_out/r5/h_comm_probe_gloo2.json; echo; grep -v Gloo gpurun_out/r5/h_comm_probe_gloo2.err | tail -3
fi
