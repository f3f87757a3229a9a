#!/bin/bash
# round 4, x: the 16x16x32 ping-pong GEMM (k_gemm_pp16, opt-in WF_GEMM_MFMA=16, round 3's DMA scheme) against the shipped 32x32x16 kernel (round 4's DMA
# scheme) in the bench window, interleaved on one box -- is the GEMM at the power limit in situ now that its K loop runs at 92-93 % of the pipe?
#   -> gpurun_out/r4/x_bench_{32,16}_{1,2}.json
mkdir -p gpurun_out/r4
for r in 1 2; do
  python bench.py --no-cpu-baseline > gpurun_out/r4/x_bench_32_$r.json 2> gpurun_out/r4/x_bench_32_$r.err
  WF_GEMM_MFMA=16 python bench.py --no-cpu-baseline > gpurun_out/r4/x_bench_16_$r.json 2> gpurun_out/r4/x_bench_16_$r.err
done
python - <<'PY'
import json
for n in ("32_1","16_1","32_2","16_2"):
    d=json.loads(open(f"gpurun_out/r4/x_bench_{n}.json").read().strip().splitlines()[-1]); print(n, round(d["value"],4), round(d["guided_step_ms"]), round(d["plain_step_ms"]))
PY
