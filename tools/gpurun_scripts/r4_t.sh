#!/bin/bash
# round 4, t: read-modify-write epilogue of k_gemm_pp with the tile's old values touched during the K loop (WF_GEMM_TOUCH lab variants:
# VARIANTS=gemmtiming,gemm_touch1,gemm_touch2,gemm_touch2_24 python tools/gemm_pp_cycles.py build): cycle shares per tile, two rounds
#   -> gpurun_out/r4/t_gemm_touch.md     (kept for the record: the variants and the WF_GEMM_TOUCH switch were removed after the negative result,
#   profiles/r4_t_gemm_touch.md; to re-run, restore them from commit 'profiles: validation at HEAD after the conv work')
mkdir -p gpurun_out/r4
for r in 1 2; do
  VARIANTS=gemmtiming,gemm_touch1,gemm_touch2,gemm_touch2_24 timeout 1500 python tools/gemm_pp_cycles.py pick 2>&1 | grep -v amdgpu.ids >> gpurun_out/r4/t_gemm_touch.md
done
cut -c1-260 gpurun_out/r4/t_gemm_touch.md
