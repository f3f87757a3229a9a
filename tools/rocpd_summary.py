"""Summarise a rocprofv3 (ROCm 7.2 rocpd sqlite) kernel trace as a per-kernel stats table (the `--stats` view):
    python tools/rocpd_summary.py gpurun_out/prof/x_results.db > profiles/summary.md"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels "
                   f"group by {name_col} order by sum(end-start) desc").fetchall()
tot = sum(r[2] for r in rows)
print(f"| kernel | calls | total ms | avg us | min us | max us | % |")
print("|---|---|---|---|---|---|---|")
for n, c, s, a, mn, mx in rows[:40]:
    n = n if len(n) < 90 else n[:87] + "..."
    print(f"| `{n}` | {c} | {s / 1e6:.2f} | {a / 1e3:.1f} | {mn / 1e3:.1f} | {mx / 1e3:.1f} | {100.0 * s / tot:.2f} |")
print(f"\ntotal kernel time {tot / 1e6:.1f} ms over {sum(r[1] for r in rows)} dispatches")
