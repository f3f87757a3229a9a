#!/bin/bash
# round 4, ab: k_conv_w4's LDS epilogue with the residual lines of the tile touched before the first group, against the previous conv.hip (lab conv_prev):
# hashes, the residual conv of conv_bench, vae_bench, interleaved   -> gpurun_out/r4/ab_*
mkdir -p gpurun_out/r4
for v in conv_prev NEW; do
  echo "== $v" >> gpurun_out/r4/ab_conv_check.txt
  if [ $v = NEW ]; then unset WF_LIB; else export WF_LIB=$PWD/worldforge_amd/_lib/lab/libwf_hip_$v.so; fi
  timeout 600 python tools/conv_check.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/r4/ab_conv_check.txt
done
for r in 1 2 3; do
  for v in conv_prev NEW; do
    echo "== $v (round $r)" >> gpurun_out/r4/ab_conv_ab.txt
    if [ $v = NEW ]; then unset WF_LIB; else export WF_LIB=$PWD/worldforge_amd/_lib/lab/libwf_hip_$v.so; fi
    timeout 600 python tools/conv_bench.py 2>/dev/null | grep -E "resid" >> gpurun_out/r4/ab_conv_ab.txt
    timeout 600 python tools/vae_bench.py 2>/dev/null | grep fp32 >> gpurun_out/r4/ab_conv_ab.txt
  done
done
python - <<'PY'
lines=open('gpurun_out/r4/ab_conv_check.txt').read().split('\n')
i=lines.index('== NEW'); a=[l for l in lines[1:i] if l]; b=[l for l in lines[i+1:] if l]
print("hashes equal:", a==b, len(a), len(b))
PY
cat gpurun_out/r4/ab_conv_ab.txt
