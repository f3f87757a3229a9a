"""rocprofv3 `*kernel_stats.csv` -> the markdown table kept under profiles/ (kernel | calls | total ms | avg us | %), kernels >= 0.01 %.
    python tools/kernel_stats_md.py gpurun_out/final/kernel_stats.csv "header text" > profiles/rN_final_kernel_stats.md"""
import csv, re, sys

rows = list(csv.DictReader(open(sys.argv[1])))
total = sum(int(r["TotalDurationNs"]) for r in rows)
print((sys.argv[2] if len(sys.argv) > 2 else "rocprofv3 --kernel-trace --stats") + "\n")
print("| kernel | calls | total ms | avg us | % |\n|---|---|---|---|---|")
for r in rows:
    pct = 100.0 * int(r["TotalDurationNs"]) / total
    if pct < 0.01:
        continue
    name = re.sub(r"\(anonymous namespace\)::", "", r["Name"]).replace("void ", "")
    name = name if len(name) <= 100 else name[:100] + "..."
    print(f"| `{name}` | {r['Calls']} | {int(r['TotalDurationNs']) / 1e6:.1f} | {float(r['AverageNs']) / 1e3:.1f} | {pct:.2f} |")
print(f"\ntotal kernel time {total / 1e6:.1f} ms")
