#!/bin/bash
# round 3, q: attention schedule lab (tools/attn_lab.py): variants + ablations of k_attn_w4<4>, optional per-8-gap-group cycle accounting
set -x
mkdir -p gpurun_out/r3
python tools/attn_lab.py run --rounds ${ROUNDS:-2} --variants ${VARIANTS:-base,place3} > gpurun_out/r3/q_attn_lab.txt 2> gpurun_out/r3/q_attn_lab.err
tail -25 gpurun_out/r3/q_attn_lab.txt
for v in ${TIMING:-}; do
  echo "== $v" >> gpurun_out/r3/q_attn_timing.txt
  WF_LIB=worldforge_amd/_lib/lab/libwf_hip_$v.so python tools/attn_timing.py >> gpurun_out/r3/q_attn_timing.txt 2>> gpurun_out/r3/q_attn_lab.err
done
