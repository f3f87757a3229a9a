#!/bin/bash
# round 6, g: the sharded VAE after dividing a row group's attention queries among its replicas and trimming the replicated layout work of
# the standalone encode / decode: VAE / multi-rank / RCCL tests, the per-rank probe, one rank of 8 in lock-step next to a 1-GPU line
#   -> gpurun_out/r6/g_*
mkdir -p gpurun_out/r6
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_vae.py tests/test_gpu_multirank.py tests/test_gpu_rccl2.py tests/test_gpu_fullsize.py tests/test_gpu_config3.py -m gpu -q 2>&1 | tail -12 > gpurun_out/r6/g_pytest.log; tail -5 gpurun_out/r6/g_pytest.log
timeout 600 python tools/vae_rank_probe.py > gpurun_out/r6/g_vae_rank_probe.txt 2>&1; tail -6 gpurun_out/r6/g_vae_rank_probe.txt
for spec in "1:--no-also" "8:--as-rank-of 8"; do
  name=${spec%%:*}; args=${spec#*:}
  timeout 600 python bench.py $args --no-cpu-baseline > gpurun_out/r6/g_asrank_$name.json 2> gpurun_out/r6/g_asrank_$name.err; echo "asrank $name rc=$?"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r6/g_asrank_*.json")):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split("/")[-1], "value", round(d.get("value"), 4), "calib", (d.get("box_calib_tflops") or {}).get("mean"), "g/p ms", d.get("guided_step_ms"), d.get("plain_step_ms"))
PY
