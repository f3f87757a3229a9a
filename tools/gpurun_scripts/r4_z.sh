#!/bin/bash
# round 4, z: does breaking the lockstep of k_gemm_pp's workgroups hide the read-modify-write epilogue?  tools/gemm_dephase.py on a lab library with
# WF_GEMM_TILE read per call (WF_EXTRA_HIPCC_FLAGS=-DWF_GEMM_LAB_TILE python tools/lab_lib.py gemm_labtile gemm.hip=WORK)   -> gpurun_out/r4/z_gemm_dephase.txt
mkdir -p gpurun_out/r4
WF_LIB=$PWD/worldforge_amd/_lib/lab/libwf_hip_gemm_labtile.so timeout 900 python tools/gemm_dephase.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4/z_gemm_dephase.txt
cat gpurun_out/r4/z_gemm_dephase.txt
