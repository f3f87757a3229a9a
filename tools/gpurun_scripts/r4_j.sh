#!/bin/bash
# round 4, j: k_conv_w4 with the round-4 address paths (running weight pointer + immediate offsets, per-slice LDS base registers, immediate M0
# for the LDS-DMA pieces) against the previous conv.hip (lab library conv_old): conv / VAE tests, conv_bench and vae_bench A/B on one box
#   -> gpurun_out/r4/j_*
mkdir -p gpurun_out/r4
python -m pytest tests/test_gpu_vae.py tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | grep -E "passed|failed|rror" | tail -5 > gpurun_out/r4/j_pytest.log
for r in 1 2; do
  for v in conv_old NEW; do
    echo "== $v (round $r)" >> gpurun_out/r4/j_conv_ab.txt
    if [ $v = NEW ]; then unset WF_LIB; else export WF_LIB=$PWD/worldforge_amd/_lib/lab/libwf_hip_$v.so; fi
    timeout 600 python tools/conv_bench.py 2>/dev/null | grep -E "slice-major|resid" >> gpurun_out/r4/j_conv_ab.txt
    timeout 600 python tools/vae_bench.py 2>/dev/null >> gpurun_out/r4/j_conv_ab.txt
  done
done
cat gpurun_out/r4/j_pytest.log; cat gpurun_out/r4/j_conv_ab.txt
