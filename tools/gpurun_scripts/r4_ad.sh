#!/bin/bash
# round 4, ad: the VAE mid-block's P . V products of all frames as ONE batched launch (wf_gemm_f16_batched) against the per-frame launches
# (WF_VAE_ATTN_UNBATCHED=1, same library): VAE / full-size / multi-rank tests, vae_bench interleaved, one simulated rank of 8 both ways
#   -> gpurun_out/r4/ad_*
mkdir -p gpurun_out/r4
python -m pytest tests/test_gpu_vae.py tests/test_gpu_fullsize.py tests/test_gpu_multirank.py -m gpu -q -x 2>&1 | grep -E "passed|failed|rror" | tail -5 > gpurun_out/r4/ad_pytest.log
for r in 1 2; do
  for v in unbatched batched; do
    echo "== $v (round $r)" >> gpurun_out/r4/ad_vae_ab.txt
    if [ $v = batched ]; then unset WF_VAE_ATTN_UNBATCHED; else export WF_VAE_ATTN_UNBATCHED=1; fi
    timeout 600 python tools/vae_bench.py 2>/dev/null | grep fp32 >> gpurun_out/r4/ad_vae_ab.txt
  done
done
for v in unbatched batched; do
  if [ $v = batched ]; then unset WF_VAE_ATTN_UNBATCHED; else export WF_VAE_ATTN_UNBATCHED=1; fi
  python bench.py --no-cpu-baseline --as-rank-of 8 > gpurun_out/r4/ad_asrank8_$v.json 2> gpurun_out/r4/ad_asrank8_$v.err
done
unset WF_VAE_ATTN_UNBATCHED
cat gpurun_out/r4/ad_pytest.log gpurun_out/r4/ad_vae_ab.txt
python - <<'PY'
import json
for v in ("unbatched","batched"):
    d=json.loads(open(f"gpurun_out/r4/ad_asrank8_{v}.json").read().strip().splitlines()[-1]); print(v, round(d["value"],4), round(d["guided_step_ms"],1), round(d["plain_step_ms"],1))
PY
