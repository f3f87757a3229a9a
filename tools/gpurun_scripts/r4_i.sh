#!/bin/bash
# round 4, i: the bench window with the GEMM's LDS-DMA pieces as in rounds 1-3 (lab library gemm_dma_in_mfma) and as at HEAD (gemmtiming:
# saddr form, 5 pieces in the read phase), same box, two interleaved rounds -> gpurun_out/r4/i_*
mkdir -p gpurun_out/r4
for r in 1 2; do
  for v in gemm_dma_in_mfma gemmtiming; do
    WF_LIB=worldforge_amd/_lib/lab/libwf_hip_$v.so python bench.py --no-cpu-baseline > gpurun_out/r4/i_bench_${v}_$r.json 2> gpurun_out/r4/i_bench_${v}_$r.err
    python - <<PY
import json
d = json.load(open("gpurun_out/r4/i_bench_${v}_$r.json"))
print("$v round $r:", round(d["value"], 4), "steps/s  guided", round(d["guided_step_ms"]), "plain", round(d["plain_step_ms"]), "attn ms", round(d["roofline"]["avg_launch_ms"], 3))
PY
  done
done
