#!/bin/bash
# round 4, d: the fp16 three-term VAE ("fp16x3", the new fp32-class default) -- VAE / sampler / e2e tests with the tolerance log, and the
# bench window with fp16x3 vs bf16x3 on one box -> gpurun_out/r4/d_*
mkdir -p gpurun_out/r4
rm -f gpurun_out/r4/d_tolerances.txt
WF_TOL_LOG=$PWD/gpurun_out/r4/d_tolerances.txt python -m pytest tests/test_gpu_vae.py tests/test_gpu_multirank.py tests/test_gpu_schedule_length.py tests/test_gpu_e2e.py tests/test_config1_truck.py tests/test_gpu_fullsize.py tests/test_gpu_timed_kernel_parity.py tests/test_gpu_bsa.py tests/test_infer_entry.py -m gpu -q -x -s 2>&1 | grep -v "^$" | tail -80 > gpurun_out/r4/d_pytest.log
python bench.py --no-cpu-baseline > gpurun_out/r4/d_bench_fp16x3.json 2> gpurun_out/r4/d_bench_fp16x3.err
python bench.py --no-cpu-baseline --vae-precision bf16x3 > gpurun_out/r4/d_bench_bf16x3.json 2> gpurun_out/r4/d_bench_bf16x3.err
tail -40 gpurun_out/r4/d_pytest.log; head -c 400 gpurun_out/r4/d_bench_fp16x3.json; echo; head -c 400 gpurun_out/r4/d_bench_bf16x3.json
