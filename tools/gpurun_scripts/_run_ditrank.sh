#!/bin/bash
mkdir -p gpurun_out/ditrank
export TMPDIR=/tmp
P=8 LAYERS=4 N=2 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ditrank/prof -o t -- python3 tools/dit_rank_probe.py > gpurun_out/ditrank/out.txt 2>&1
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/ditrank/prof/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
# last forward pair only: take the last half of the rows by start time
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
half = rows[len(rows) // 2:]
agg = collections.OrderedDict()
for r in half:
    n = r['Kernel_Name'][:70]
    key = (n, r.get('Grid_Size_X') or r.get('Grid_Size'), r.get('Workgroup_Size_X') or r.get('Workgroup_Size'))
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    a = agg.setdefault(key, [0, 0.0])
    a[0] += 1; a[1] += d
tot = sum(v[1] for v in agg.values())
print("second pair: kernel time %.1f ms" % (tot / 1e3))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%8.1f us x%4d  grid %s wg %s  %s" % (v[1] / v[0], v[0], k[1], k[2], k[0]))
PY
find gpurun_out/ditrank/prof -type f -delete
