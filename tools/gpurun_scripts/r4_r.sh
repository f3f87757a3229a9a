#!/bin/bash
# round 4, r: k_conv_w4 with the LDS-transposed epilogue against the previous conv.hip (lab library conv_prev): output hashes, VAE / full-size
# tests, conv_bench / vae_bench A/B, epilogue cycles on the instrumented library   -> gpurun_out/r4/r_*
mkdir -p gpurun_out/r4
for v in conv_prev NEW; do
  echo "== $v" >> gpurun_out/r4/r_conv_check.txt
  if [ $v = NEW ]; then unset WF_LIB; else export WF_LIB=$PWD/worldforge_amd/_lib/lab/libwf_hip_$v.so; fi
  timeout 600 python tools/conv_check.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r4/r_conv_check.txt
done
unset WF_LIB
python -m pytest tests/test_gpu_vae.py tests/test_gpu_fullsize.py tests/test_gpu_multirank.py -m gpu -q -x 2>&1 | grep -E "passed|failed|rror" | tail -5 > gpurun_out/r4/r_pytest.log
for r in 1 2; do
  for v in conv_prev NEW; do
    echo "== $v (round $r)" >> gpurun_out/r4/r_conv_ab.txt
    if [ $v = NEW ]; then unset WF_LIB; else export WF_LIB=$PWD/worldforge_amd/_lib/lab/libwf_hip_$v.so; fi
    timeout 600 python tools/conv_bench.py 2>/dev/null | grep -E "slice-major|resid" >> gpurun_out/r4/r_conv_ab.txt
    timeout 600 python tools/vae_bench.py 2>/dev/null >> gpurun_out/r4/r_conv_ab.txt
  done
done
export WF_LIB=$PWD/worldforge_amd/_lib/lab/libwf_hip_convtiming.so
WF_CONV_DEBUG=0 X3=1 timeout 600 python tools/conv_timing.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r4/r_conv_cycles.txt
python - <<'PY'
lines=open('gpurun_out/r4/r_conv_check.txt').read().split('\n')
i=lines.index('== NEW'); a=[l for l in lines[1:i] if l]; b=[l for l in lines[i+1:] if l]
print("hashes equal:", a==b, len(a), len(b))
for x,y in zip(a,b):
    if x!=y: print(x); print(y)
PY
cat gpurun_out/r4/r_pytest.log; cat gpurun_out/r4/r_conv_ab.txt; cut -c1-200 gpurun_out/r4/r_conv_cycles.txt
