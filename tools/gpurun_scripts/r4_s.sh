#!/bin/bash
# round 4, s: epilogue cycles of k_conv_w4 (instrumented lab library, bias + fp32 out = the VAE's common epilogue) -> gpurun_out/r4/s_conv_cycles.txt
mkdir -p gpurun_out/r4
export WF_LIB=$PWD/worldforge_amd/_lib/lab/libwf_hip_convtiming.so
WF_CONV_DEBUG=0 X3=1 timeout 600 python tools/conv_timing.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4/s_conv_cycles.txt
cut -c1-200 gpurun_out/r4/s_conv_cycles.txt
