#!/bin/bash
# the documented A/B switches still give a working (and parity-green) engine
mkdir -p gpurun_out
: > gpurun_out/switches.txt
for sw in "WF_CONV_WALK=0" "WF_VAE_UP2_PHASES=0" "WF_ATTN_PRESCALE=0" "WF_ATTN_TRACK_MAX=1" "WF_BSA_TORCH_SELECT=1" "WF_CTX_REPLICATED=1 WF_VAE_LOWRES_REPLICATED=1" "WF_ATTN_KERNEL=w8" "WF_CONV_NO_W4=1"; do
  echo "== $sw" >> gpurun_out/switches.txt
  env $sw python -m pytest tests/test_gpu_vae.py tests/test_gpu_dit.py tests/test_gpu_e2e.py tests/test_gpu_multirank.py tests/test_gpu_bsa.py -q -x 2>&1 | grep -E "passed|failed|rror" | tail -2 >> gpurun_out/switches.txt
done
cat gpurun_out/switches.txt
