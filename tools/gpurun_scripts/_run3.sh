set -x
python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_schedule_length.py > gpurun_out/r2_pytest_c.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_pytest_c.log
tail -5 gpurun_out/r2_pytest_c.log
timeout 900 python bench.py > gpurun_out/r2_bench_b_fp32vae.json 2> gpurun_out/r2_bench_b_fp32vae.err; echo "rc=$?"
timeout 900 python bench.py --vae-precision bf16 --no-cpu-baseline > gpurun_out/r2_bench_b_bf16vae.json 2> gpurun_out/r2_bench_b_bf16vae.err; echo "rc=$?"
cat gpurun_out/r2_bench_b_fp32vae.json gpurun_out/r2_bench_b_bf16vae.json
cd /tmp && export TMPDIR=/tmp
L=4096 N=1 timeout 240 rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $GRAFT_REPO_ROOT/gpurun_out/r2_pmc_small -o pmc -- python3 $GRAFT_REPO_ROOT/tools/attn_once.py > $GRAFT_REPO_ROOT/gpurun_out/r2_pmc_small.log 2>&1; echo "pmc small rc=$?"
tail -5 $GRAFT_REPO_ROOT/gpurun_out/r2_pmc_small.log
ls -R $GRAFT_REPO_ROOT/gpurun_out/r2_pmc_small | head
