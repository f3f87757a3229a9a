set -x
mkdir -p gpurun_out/r3
WF_GEMM_MFMA=16 python -m pytest tests/test_gpu_dit.py -m gpu -q -k "test_gemm and not equals_small" > gpurun_out/r3/i_gemm16_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r3/i_gemm16_tests.log; tail -6 gpurun_out/r3/i_gemm16_tests.log
WF_GEMM_MFMA=16 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -k "gemm" >> gpurun_out/r3/i_gemm16_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r3/i_gemm16_tests.log; tail -4 gpurun_out/r3/i_gemm16_tests.log
python tools/gemm_tiles.py > gpurun_out/r3/i_gemm_tiles_mfma32.txt 2>&1
WF_GEMM_MFMA=16 python tools/gemm_tiles.py > gpurun_out/r3/i_gemm_tiles_mfma16.txt 2>&1
paste -d'|' <(grep "^P=" gpurun_out/r3/i_gemm_tiles_mfma32.txt | tail -20) <(grep "^P=" gpurun_out/r3/i_gemm_tiles_mfma16.txt | tail -20 | awk '{print $(NF-3), $(NF-2), $(NF-1), $NF}')
python bench.py --no-cpu-baseline > gpurun_out/r3/i_bench32.json 2>/dev/null
WF_GEMM_MFMA=16 python bench.py --no-cpu-baseline > gpurun_out/r3/i_bench16.json 2>/dev/null
python - <<'PY'
import json
for n in ("i_bench32","i_bench16"):
    d=json.load(open(f"gpurun_out/r3/{n}.json")); print(n, round(d["value"],4), round(d["guided_step_ms"]), round(d["plain_step_ms"]), round(d["roofline"]["achieved"]))
PY
