set -x
mkdir -p gpurun_out/r3
python tools/conv_bench.py > gpurun_out/r3/f_conv_bench.txt 2>&1; cat gpurun_out/r3/f_conv_bench.txt
python -m pytest tests/test_gpu_vae.py tests/test_gpu_fullsize.py tests/test_gpu_config3.py -m gpu -q -x --durations=5 > gpurun_out/r3/f_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r3/f_tests.log
tail -12 gpurun_out/r3/f_tests.log
python tools/vae_bench.py > gpurun_out/r3/f_vae_bench.txt 2>&1; cat gpurun_out/r3/f_vae_bench.txt
python bench.py --no-cpu-baseline > gpurun_out/r3/f_bench.json 2> gpurun_out/r3/f_bench.err; echo "rc=$?"
python -c "
import json; d=json.load(open('gpurun_out/r3/f_bench.json')); print(round(d['value'],4), round(d['guided_step_ms']), round(d['plain_step_ms']), d['roofline']['achieved'])"
