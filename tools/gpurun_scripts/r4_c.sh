#!/bin/bash
# round 4, c: (1) does the fp16 MFMA honour subnormal inputs (tools/gemm_lab/f16_denorm.hip), (2) cycle shares of k_gemm_pp per output tile
# (tools/gemm_pp_cycles.py on a -DWF_GEMM_TIMING build) -> gpurun_out/r4/c_*
mkdir -p gpurun_out/r4
timeout 60 tools/gemm_lab/f16_denorm > gpurun_out/r4/c_f16_denorm.txt 2>&1
timeout 600 python tools/gemm_pp_cycles.py run > gpurun_out/r4/c_gemm_pp_cycles.md 2> gpurun_out/r4/c_gemm_pp_cycles.err
cat gpurun_out/r4/c_f16_denorm.txt; cat gpurun_out/r4/c_gemm_pp_cycles.md; tail -3 gpurun_out/r4/c_gemm_pp_cycles.err
