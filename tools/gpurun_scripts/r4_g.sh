#!/bin/bash
# round 4, g: k_gemm_pp with the LDS-DMA pieces issued in the wave's own READ phase (WF_GEMM_DMA_PHASE=1, the new default) against the
# in-MFMA-phase placement of rounds 1-3: GEMM / DiT tests, cycle table of every DiT shape, same-box A/B -> gpurun_out/r4/g_*
mkdir -p gpurun_out/r4
python -m pytest tests/test_gpu_dit.py tests/test_gpu_fullsize.py tests/test_gpu_vae.py -m gpu -q -x 2>&1 | tail -5 > gpurun_out/r4/g_pytest.log
timeout 600 python tools/gemm_pp_cycles.py run > gpurun_out/r4/g_gemm_cycles_new.md 2> gpurun_out/r4/g_err.txt
for v in gemm_dma_in_mfma gemmtiming gemm_dma_in_mfma gemmtiming; do
  echo "== $v" >> gpurun_out/r4/g_gemm_ab.md
  WF_LIB=worldforge_amd/_lib/lab/libwf_hip_$v.so timeout 300 python tools/gemm_pp_cycles.py child >> gpurun_out/r4/g_gemm_ab.md 2>> gpurun_out/r4/g_err.txt
done
tail -3 gpurun_out/r4/g_pytest.log; cat gpurun_out/r4/g_gemm_ab.md | grep -v "^|---" | cut -c1-250
