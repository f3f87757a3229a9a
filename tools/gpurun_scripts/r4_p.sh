#!/bin/bash
# round 4, p: k_conv_w4 with the output-block-major tap order (weight fragments reloaded three taps ahead right after their last MFMA) against
# the previous conv.hip (lab library: python tools/lab_lib.py conv_prev conv.hip=<rev>): output hashes, VAE tests, conv_bench / vae_bench A/B
#   -> gpurun_out/r4/p_*
mkdir -p gpurun_out/r4
for v in conv_prev NEW; do
  echo "== $v" >> gpurun_out/r4/p_conv_check.txt
  if [ $v = NEW ]; then unset WF_LIB; else export WF_LIB=$PWD/worldforge_amd/_lib/lab/libwf_hip_$v.so; fi
  timeout 600 python tools/conv_check.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r4/p_conv_check.txt
done
unset WF_LIB
python -m pytest tests/test_gpu_vae.py tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | grep -E "passed|failed|rror" | tail -5 > gpurun_out/r4/p_pytest.log
for r in 1 2; do
  for v in conv_prev NEW; do
    echo "== $v (round $r)" >> gpurun_out/r4/p_conv_ab.txt
    if [ $v = NEW ]; then unset WF_LIB; else export WF_LIB=$PWD/worldforge_amd/_lib/lab/libwf_hip_$v.so; fi
    timeout 600 python tools/conv_bench.py 2>/dev/null | grep -E "slice-major|resid" >> gpurun_out/r4/p_conv_ab.txt
    timeout 600 python tools/vae_bench.py 2>/dev/null >> gpurun_out/r4/p_conv_ab.txt
  done
done
cat gpurun_out/r4/p_conv_check.txt | awk '{print $NF, $0}' | cut -c1-120 | sort | uniq -c | sort -rn | head -3; cat gpurun_out/r4/p_pytest.log; cat gpurun_out/r4/p_conv_ab.txt
