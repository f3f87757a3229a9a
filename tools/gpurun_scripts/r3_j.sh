set -x
R=$GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3/j_prof32 -- python3 $R/bench.py --no-cpu-baseline --steps 4 > $R/gpurun_out/r3/j_bench32.json 2> /dev/null
export WF_GEMM_MFMA=16
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3/j_prof16 -- python3 $R/bench.py --no-cpu-baseline --steps 4 > $R/gpurun_out/r3/j_bench16.json 2> /dev/null
cd $R
for d in j_prof32 j_prof16; do find gpurun_out/r3/$d -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r3/${d}_kernel_stats.csv; rm -rf gpurun_out/r3/$d; done
python - <<'PY'
import csv, re, json
def load(p):
    d={}
    for r in csv.DictReader(open(p)):
        n=re.sub(r'\(anonymous namespace\)::','',r['Name']); n=re.sub(r'^void ','',n)[:40]
        d[n]=(int(r['Calls']), float(r['TotalDurationNs'])/1e6)
    return d
a,b=load('gpurun_out/r3/j_prof32_kernel_stats.csv'),load('gpurun_out/r3/j_prof16_kernel_stats.csv')
for k in sorted(set(a)|set(b), key=lambda k:-(a.get(k,(0,0))[1]+b.get(k,(0,0))[1]))[:14]:
    print(f"{k:42s} mfma32 {a.get(k,(0,0))[1]:9.1f} ms ({a.get(k,(0,0))[0]})   mfma16 {b.get(k,(0,0))[1]:9.1f} ms ({b.get(k,(0,0))[0]})")
print('total', sum(v[1] for v in a.values()), sum(v[1] for v in b.values()))
for n in ('j_bench32','j_bench16'):
    d=json.load(open(f'gpurun_out/r3/{n}.json')); print(n, d['value'], d['plain_step_ms'], d['guided_step_ms'], d['roofline']['achieved'])
PY
