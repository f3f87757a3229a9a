#!/bin/bash
# round 5, l (as k, plus the up- / down-sampling convs border-first): the border rows of a slab first, exchanged on the communication stream while the norm kernel produces the slab (vae._halo_operand) -- the sharded-VAE
# tests, then one rank of 8 communication-free and under the 330 / 165 GB/s bandwidth models (r5_j_*: the neighbour exchange synchronous)
#   -> gpurun_out/r5/l_*
mkdir -p gpurun_out/r5
timeout 1500 python -m pytest tests/test_gpu_vae.py tests/test_gpu_multirank.py tests/test_gpu_rccl2.py tests/test_gpu_config3.py tests/test_gpu_longcat_sampler.py -m gpu -q 2>&1 | grep -E "passed|failed|^FAILED" | tail -4
timeout 600 python bench.py --as-rank-of 8 --no-cpu-baseline > gpurun_out/r5/l_asrank8.json 2> gpurun_out/r5/l_asrank8.err; echo "asrank8 rc=$?"
for spec in "ag330:330,153,20" "ag165:165,153,30"; do
  name=${spec%%:*}; m=${spec#*:}
  timeout 600 python bench.py --as-rank-of 8 --emulate-comm $m --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r5/l_wan_$name.json 2> gpurun_out/r5/l_wan_$name.err; echo "wan $name rc=$?"
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r5/l_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "value", round(d.get("value"), 4), "g/p ms", round(d.get("guided_step_ms") or 0), round(d.get("plain_step_ms") or 0, 1), "selected", (d.get("exchange") or {}).get("selected"), "calib", (d.get("box_calib_tflops") or {}).get("mean"))
    except Exception as e:
        print(f, "unreadable", e)
PY
