"""Summarise rocprofv3 --pmc passes (csv output: *_counter_collection.csv) per kernel: mean counter value per dispatch.

    python tools/pmc_summary.py <dir> [<dir> ...] [--kernel SUBSTR]
    python tools/pmc_summary.py <dir> ... --kernel "k_attn_w4<4>" --attn-json profiles/attn_pmc_latest.json --tokens 32760 --heads 40 \
           --source "profiles/r6_attn_pmc_summary.txt"       # the roofline side fields bench.py reads (warm = last dispatch of each pass)
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


def attn_json(acc, out, tokens, heads, source, kernel_label):
    """bench.py's roofline side fields from the warm (last) dispatch of each counter pass.  gfx950 corrections as MI355X_MICROARCH.md
    prescribes: FETCH_SIZE (KB) x 2 for wide coalesced reads; effective clock = GRBM_GUI_ACTIVE per XCD / kernel duration is not available
    from the counter csv alone, so the clock is derived from SQ_BUSY_CYCLES-free quantities: cycles per XCD = GRBM_GUI_ACTIVE / 8 and the
    MFMA-pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / that."""
    import json
    c = {}
    for k, cs in acc.items():
        for cn, vs in cs.items():
            c[cn] = vs[-1]
    cyc_xcd = c["GRBM_GUI_ACTIVE"] / 8.0
    res = {"kernel": kernel_label, "tokens": tokens, "heads": heads, "source": source,
           "fetch_size_kb": c.get("FETCH_SIZE"), "write_size_kb": c.get("WRITE_SIZE"),
           "traffic_bytes_per_launch": int(c["FETCH_SIZE"] * 1024 * 2 + c["WRITE_SIZE"] * 1024) if "FETCH_SIZE" in c and "WRITE_SIZE" in c else None,
           "mfma_util": c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / cyc_xcd, "cycles_per_xcd": cyc_xcd,
           "l2_hit_rate": (c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])) if "TCC_HIT_sum" in c else None,
           "lds_bank_conflict_cycles": c.get("SQ_LDS_BANK_CONFLICT")}
    dur_ms = float(os.environ["PMC_DURATION_MS"]) if os.environ.get("PMC_DURATION_MS") else None   # kernel-trace duration of the warm launch in the mfma pass
    if dur_ms:
        res["duration_ms_profiled"] = dur_ms
        res["clock_ghz"] = cyc_xcd / (dur_ms * 1e-3) / 1e9
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))


def _opt(name):
    if name in sys.argv:
        v = sys.argv[sys.argv.index(name) + 1]
        return v
    return None


if __name__ == "__main__":
    opts = {n: _opt(n) for n in ("--kernel", "--attn-json", "--tokens", "--heads", "--source")}
    skip = set(v for v in opts.values() if v is not None)
    args = [a for a in sys.argv[1:] if not a.startswith("--") and a not in skip]
    kern = opts["--kernel"]
    acc = summarise(args, kern)
    if opts["--attn-json"]:
        attn_json(acc, opts["--attn-json"], int(opts["--tokens"] or 32760), int(opts["--heads"] or 40), opts["--source"] or "", kern)
    for k, cs in acc.items():
        print(f"kernel: {k[:120]}")
        for cn, vs in sorted(cs.items()):
            print(f"  {cn:32s} n={len(vs):3d} mean={sum(vs) / len(vs):.6g}")
