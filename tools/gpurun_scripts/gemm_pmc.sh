#!/bin/bash
# rocprofv3 PMC pass of the ping-pong GEMM on the DiT's QKV and O-projection shapes (tools/gemm_once.py, random operands): effective clock and MFMA-pipe busy
# (round 2's table: profiles/r2_gemm_clock_pmc.md).   gpurun --timeout 900 -- 'bash tools/gpurun_scripts/gemm_pmc.sh'   -> gpurun_out/gemm_pmc/summary.txt
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/gemm_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for shape in qkv o ffn_down; do
  SHAPE=$shape ZERO=0 N=3 timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY -d $OUT/$shape -o pmc -- python3 $R/tools/gemm_once.py > $OUT/$shape.log 2>&1; echo "pmc $shape rc=$?"
done
cd $R
python - <<'PY' > gpurun_out/gemm_pmc/summary.txt
import csv, glob
for shape in ("qkv", "o", "ffn_down"):
    dur = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6 for f in glob.glob(f'gpurun_out/gemm_pmc/{shape}/**/*kernel_trace.csv', recursive=True) for r in csv.DictReader(open(f)) if 'k_gemm_pp' in r['Kernel_Name']]
    acc = {}
    for f in glob.glob(f'gpurun_out/gemm_pmc/{shape}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'k_gemm_pp' in r['Kernel_Name']:
                acc.setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
    print(shape, "durations ms", [round(d, 3) for d in dur])
    for k, v in sorted(acc.items()):
        print(f"  {k:28s} n={len(v)} last={v[-1]:.5g}")
    if dur and 'GRBM_GUI_ACTIVE' in acc:
        cyc = acc['GRBM_GUI_ACTIVE'][-1] / 8
        print(f"  -> {cyc / 1e6:.3f} M cycles per XCD, effective clock {cyc / dur[-1] / 1e6:.2f} GHz, MFMA pipe busy {100 * acc['SQ_VALU_MFMA_BUSY_CYCLES'][-1] / 1024 / cyc:.1f} %, waits {100 * acc['SQ_WAIT_ANY'][-1] / acc['SQ_WAVE_CYCLES'][-1]:.1f} % of the wave cycles")
PY
cat gpurun_out/gemm_pmc/summary.txt
find $OUT -type f -size +2M -delete
