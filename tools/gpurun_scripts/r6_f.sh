#!/bin/bash
# round 6, f: the part launches on their own kernel instantiation (both sides of the hole as one sequence, merge in the last launch's epilogue):
# exchange / multi-rank / DiT / LongCat tests, then the per-rank compute cost of the exchange modes on ONE box (lock-step, chunked2, chunked1,
# gather; LongCat distilled), and the RGB blend kernel
#   -> gpurun_out/r6/f_*
mkdir -p gpurun_out/r6
R=$GRAFT_REPO_ROOT
rm -f gpurun_out/r6/f_tolerances.txt
WF_TOL_LOG=$R/gpurun_out/r6/f_tolerances.txt timeout 1500 python -m pytest tests/test_gpu_exchange.py tests/test_gpu_multirank.py tests/test_gpu_dit.py tests/test_gpu_longcat.py tests/test_gpu_rccl2.py tests/test_gpu_inject_kernels.py tests/test_gpu_config3.py tests/test_gpu_fullsize.py tests/test_gpu_timed_kernel_parity.py -m gpu -q 2>&1 | tail -25 > gpurun_out/r6/f_pytest.log; tail -6 gpurun_out/r6/f_pytest.log
for spec in "8:--as-rank-of 8" "8c2:--as-rank-of 8 --exchange chunked2" "8c1:--as-rank-of 8 --exchange chunked1" "8g:--as-rank-of 8 --exchange gather"; do
  name=${spec%%:*}; args=${spec#*:}
  timeout 600 python bench.py $args --no-cpu-baseline > gpurun_out/r6/f_asrank_$name.json 2> gpurun_out/r6/f_asrank_$name.err; echo "asrank $name rc=$?"
done
timeout 600 python bench.py --workload longcat --distill --as-rank-of 8 --steps 4 --no-cpu-baseline > gpurun_out/r6/f_longcat_asrank8.json 2> gpurun_out/r6/f_longcat_asrank8.err; echo "longcat rc=$?"
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r6/f_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], "value", round(d.get("value"), 4), "calib", (d.get("box_calib_tflops") or {}).get("mean"), "g/p ms", d.get("guided_step_ms"), d.get("plain_step_ms"), "attn", (d.get("roofline") or {}).get("avg_launch_ms"), "hbm", (d.get("hbm") or {}).get("frac_of_8TBps"))
        if d.get("exchange"): print("   exchange", json.dumps(d["exchange"].get("estimated_ms_per_evaluation"))[:600])
    except Exception as e:
        print(f, "unreadable", e)
PY
