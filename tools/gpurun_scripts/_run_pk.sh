#!/bin/bash
mkdir -p gpurun_out
( echo "== new (v_pk_add_f32 row sums)"; python tools/attn_bench.py; echo "== old"; WF_LIB=$PWD/worldforge_amd/_lib/libwf_hip_old.so python tools/attn_bench.py; echo "== new again"; python tools/attn_bench.py ) > gpurun_out/pk_ab.txt 2>&1
python -m pytest tests/test_gpu_dit.py tests/test_gpu_fullsize.py tests/test_gpu_attention.py -q -x 2>&1 | tail -5 >> gpurun_out/pk_ab.txt
cat gpurun_out/pk_ab.txt
