"""Summarise rocprofv3 --pmc passes (csv output: *_counter_collection.csv) per kernel: mean counter value per dispatch.

    python tools/pmc_summary.py <dir> [<dir> ...] [--kernel SUBSTR]
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def summarise(dirs, kernel=None):
    acc = defaultdict(lambda: defaultdict(list))
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            per_dispatch = defaultdict(float)
            names = {}
            with open(f) as fh:
                for r in csv.DictReader(fh):
                    k = r.get("Kernel_Name", "")
                    if kernel and kernel not in k:
                        continue
                    key = (r.get("Dispatch_Id"), r["Counter_Name"])
                    per_dispatch[key] += float(r["Counter_Value"])
                    names[r.get("Dispatch_Id")] = k
            for (did, cn), v in per_dispatch.items():
                acc[names[did]][cn].append(v)
    return acc


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    kern = None
    if "--kernel" in sys.argv:
        kern = sys.argv[sys.argv.index("--kernel") + 1]
        args = [a for a in args if a != kern]
    acc = summarise(args, kern)
    for k, cs in acc.items():
        print(f"kernel: {k[:120]}")
        for cn, vs in sorted(cs.items()):
            print(f"  {cn:32s} n={len(vs):3d} mean={sum(vs) / len(vs):.6g}")
