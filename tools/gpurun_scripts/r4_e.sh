#!/bin/bash
# round 4, e: the segmented K / V^T exchange (per-source broadcasts, attention in parts, own shard first): multi-rank / RCCL / DiT / LongCat
# tests, and one simulated rank of 8 of the LongCat workload with the segmented exchange on and off (what the parts + merge cost in compute)
#   -> gpurun_out/r4/e_*
mkdir -p gpurun_out/r4
rm -f gpurun_out/r4/e_tolerances.txt
WF_TOL_LOG=$PWD/gpurun_out/r4/e_tolerances.txt python -m pytest tests/test_gpu_multirank.py tests/test_gpu_rccl2.py tests/test_gpu_dit.py tests/test_gpu_longcat.py tests/test_gpu_longcat_sampler.py tests/test_gpu_timed_kernel_parity.py tests/test_gpu_config3.py -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r4/e_pytest.log
python bench.py --workload longcat --distill --as-rank-of 8 --steps 4 --no-cpu-baseline > gpurun_out/r4/e_longcat_asrank8_segmented.json 2> gpurun_out/r4/e_longcat_asrank8_segmented.err
WF_ATTN_SEGMENTED=0 python bench.py --workload longcat --distill --as-rank-of 8 --steps 4 --no-cpu-baseline > gpurun_out/r4/e_longcat_asrank8_one_event.json 2> gpurun_out/r4/e_longcat_asrank8_one_event.err
tail -6 gpurun_out/r4/e_pytest.log; head -c 900 gpurun_out/r4/e_longcat_asrank8_segmented.json; echo; head -c 900 gpurun_out/r4/e_longcat_asrank8_one_event.json; echo; tail -3 gpurun_out/r4/e_longcat_asrank8_segmented.err
