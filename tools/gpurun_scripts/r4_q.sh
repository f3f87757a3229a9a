#!/bin/bash
# round 4, q: cycle-counted ablation of k_conv_w4's slice loop at HEAD (lab library: WF_EXTRA_HIPCC_FLAGS="-DWF_CONV_TIMING -DWF_CONV_ABLATE"
# python tools/lab_lib.py convtiming conv.hip=WORK): shipped, no LDS-DMA pieces (1), no weight loads (2), neither (3), no LDS reads (4), none (7)
#   -> gpurun_out/r4/q_conv_cycles_ablation.txt   (round 3's table: profiles/r3_x_conv_cycles_ablation.txt)
mkdir -p gpurun_out/r4
export WF_LIB=$PWD/worldforge_amd/_lib/lab/libwf_hip_convtiming.so
for d in 0 1 2 3 4 7; do
  echo "== WF_CONV_DEBUG=$d" >> gpurun_out/r4/q_conv_cycles_ablation.txt
  WF_CONV_DEBUG=$d X3=1 timeout 600 python tools/conv_timing.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r4/q_conv_cycles_ablation.txt
done
cat gpurun_out/r4/q_conv_cycles_ablation.txt
