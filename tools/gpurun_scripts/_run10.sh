set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
# GEMM clock evidence: random vs zero operands, same binary (QKV shape)
for z in 0 1; do
  SHAPE=qkv ZERO=$z N=3 timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES -d $R/gpurun_out/r2_pmc_gemm/zero$z -o pmc -- python3 $R/tools/gemm_once.py > $R/gpurun_out/r2_pmc_gemm_zero$z.log 2>&1; echo "gemm pmc zero=$z rc=$?"
done
# default bench under the kernel-trace profiler
timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2_prof_bench -o prof -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/r2_prof_bench.json 2> $R/gpurun_out/r2_prof_bench.err; echo "prof bench rc=$?"
# config 3 (720p) bench line, profiled
timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r2_prof_c3 -o prof -- python3 $R/bench.py --height 720 --width 1280 --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r2_prof_c3.json 2> $R/gpurun_out/r2_prof_c3.err; echo "prof c3 rc=$?"
cd $R
cat gpurun_out/r2_prof_bench.json gpurun_out/r2_prof_c3.json
ls gpurun_out/r2_prof_bench gpurun_out/r2_prof_c3 gpurun_out/r2_pmc_gemm/*
python - <<'PY'
import csv,glob,collections
for z in (0,1):
    d=f'gpurun_out/r2_pmc_gemm/zero{z}'
    kt=[r for r in csv.DictReader(open(glob.glob(d+'/*kernel_trace.csv')[0])) if 'k_gemm' in r['Kernel_Name']]
    did=kt[-1]['Dispatch_Id']; dur=(int(kt[-1]['End_Timestamp'])-int(kt[-1]['Start_Timestamp']))/1e6
    v=collections.defaultdict(float)
    for r in csv.DictReader(open(glob.glob(d+'/*counter_collection.csv')[0])):
        if r['Dispatch_Id']==did: v[r['Counter_Name']]+=float(r['Counter_Value'])
    cyc=v['GRBM_GUI_ACTIVE']/8
    print('zero' if z else 'random', kt[-1]['Kernel_Name'][:60], 'ms',dur,'TF', 2*32760*15360*5120/dur/1e9, 'clock GHz', cyc/dur/1e6, 'mfma util', v['SQ_VALU_MFMA_BUSY_CYCLES']/1024/cyc, dict(v))
PY
