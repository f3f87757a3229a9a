#!/bin/bash
# round 4, h: GEMM lab -- VARIANTS="a b c" of tools/gemm_pp_cycles.py (lab libraries under worldforge_amd/_lib/lab), every DiT shape, ROUNDS
# interleaved rounds; first the GEMM / DiT tests on the main library -> gpurun_out/r4/h_*
mkdir -p gpurun_out/r4
rm -f gpurun_out/r4/h_gemm_ab.md
python -m pytest tests/test_gpu_dit.py tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | grep -E "passed|failed|Error|error" | tail -5 > gpurun_out/r4/h_pytest.log
for r in $(seq 1 ${ROUNDS:-2}); do
  for v in ${VARIANTS:-gemm_dma_in_mfma gemmtiming}; do
    echo "== $v (round $r)" >> gpurun_out/r4/h_gemm_ab.md
    WF_LIB=worldforge_amd/_lib/lab/libwf_hip_$v.so timeout 300 python tools/gemm_pp_cycles.py child ${SHORT:-} >> gpurun_out/r4/h_gemm_ab.md 2>> gpurun_out/r4/h_err.txt
  done
done
cat gpurun_out/r4/h_pytest.log; grep -v "^|---\|^| shape" gpurun_out/r4/h_gemm_ab.md | cut -c1-200
