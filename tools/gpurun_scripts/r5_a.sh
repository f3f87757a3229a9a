#!/bin/bash
# round 5, a: the packed K / V^T exchange (parallel.KVExchange: gather / chunked / bcast, own shard first, second-window part launches),
# LongCat lock-step CFG pair, context cache on by default -- the tests that see it, then the driver's own bench command (with the `also`
# windows of configs 3 and 4) and one simulated rank of 8 with the exchange calibration (= each mode's compute cost)
#   -> gpurun_out/r5/a_*
mkdir -p gpurun_out/r5
rm -f gpurun_out/r5/a_tolerances.txt
WF_TOL_LOG=$PWD/gpurun_out/r5/a_tolerances.txt timeout 1500 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_rccl2.py tests/test_gpu_dit.py tests/test_gpu_longcat.py tests/test_gpu_longcat_sampler.py tests/test_gpu_timed_kernel_parity.py tests/test_gpu_config3.py -m gpu -q -x 2>&1 | tail -25 > gpurun_out/r5/a_pytest.log
tail -8 gpurun_out/r5/a_pytest.log
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5/a_bench.json 2> gpurun_out/r5/a_bench.err; echo "bench rc=$?"
head -c 1500 gpurun_out/r5/a_bench.json; echo; tail -3 gpurun_out/r5/a_bench.err
timeout 600 python bench.py --as-rank-of 8 --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/r5/a_asrank8.json 2> gpurun_out/r5/a_asrank8.err; echo "asrank8 rc=$?"
python - <<'PY'
import json
for f in ("gpurun_out/r5/a_bench.json", "gpurun_out/r5/a_asrank8.json"):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f, d.get("value"), d.get("guided_step_ms"), d.get("plain_step_ms"), (d.get("roofline") or {}).get("frac"), d.get("exchange"), json.dumps(d.get("also"))[:1500] if d.get("also") else None, d.get("also_s"))
    except Exception as e:
        print(f, "unreadable", e)
PY
