#!/bin/bash
# round 4, ac: order / repetition robustness of the GPU suite at HEAD: the test files in REVERSE order, then the thread-interleaved tests (simulated ranks) three more times
#   -> gpurun_out/r4/ac_*.log
mkdir -p gpurun_out/r4
python -m pytest $(ls tests/test_*.py | sort -r) -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|rror|^FAILED" | tail -8 > gpurun_out/r4/ac_reverse.log
for i in 1 2 3; do
  python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_multirank.py tests/test_gpu_gemm_streams.py -m gpu -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|^FAILED" | tail -3 >> gpurun_out/r4/ac_repeat.log
done
cat gpurun_out/r4/ac_reverse.log gpurun_out/r4/ac_repeat.log
