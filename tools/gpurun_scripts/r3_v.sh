#!/bin/bash
# round 3, v: the round-start attention kernel against HEAD's on one box (whole-library A/B, 3 interleaved rounds); the cdf list kernel's tests
set -x
mkdir -p gpurun_out/r3
python tools/attn_lab.py run --rounds 3 --variants roundstart,head > gpurun_out/r3/v_attn_round_ab.txt 2> gpurun_out/r3/v.err
tail -6 gpurun_out/r3/v_attn_round_ab.txt
python -m pytest tests/test_gpu_bsa.py -x -q -m gpu > gpurun_out/r3/v_bsa_tests.log 2>&1
echo "rc=$?" >> gpurun_out/r3/v_bsa_tests.log
tail -15 gpurun_out/r3/v_bsa_tests.log
