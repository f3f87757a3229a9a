#!/bin/bash
# round 6, a: the one-term fp16 VAE mode -- its tests, the whole VAE test file, and decode / encode timings of the three modes at C2
#   -> gpurun_out/r6/a_*
mkdir -p gpurun_out/r6
rm -f gpurun_out/r6/a_tolerances.txt
WF_TOL_LOG=$PWD/gpurun_out/r6/a_tolerances.txt timeout 1200 python -m pytest tests/test_gpu_vae.py -m gpu -q -s 2>&1 | grep -E "rel L2|passed|failed|rror|^FAILED|assert" | tail -60 > gpurun_out/r6/a_pytest.log
tail -40 gpurun_out/r6/a_pytest.log
timeout 400 python tools/vae_bench.py > gpurun_out/r6/a_vae_bench.txt 2>&1; tail -8 gpurun_out/r6/a_vae_bench.txt
