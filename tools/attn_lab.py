"""Lab for schedule variants of the timed self-attention kernel (`k_attn_w4<4>`): same source, one -D per variant, one library per variant.

    python tools/attn_lab.py build            # here (no GPU): compiles attention.hip once per variant with -DWF_ATTN_LAB (only the
                                              # pre-scaled self-attention is instantiated: ~1.5 min instead of ~7) and links
                                              # worldforge_amd/_lib/lab/libwf_hip_<variant>.so from the other objects of the normal build
    python tools/attn_lab.py run [--rounds 3] # on the GPU box: every variant in its own child process (WF_LIB), interleaved rounds,
                                              # output compared with the first variant's (max |diff|, must be 0 for schedule-only changes)

Variants are `name: flags` pairs in VARIANTS; `base` is the shipped schedule.  A new schedule idea = one more -D in attention.hip + one
line here; profiles/r3_q_attn_lab.md has the round-3 table (what was tried, what the ingredients cost).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LABLIB = os.path.join(ROOT, "worldforge_amd", "_lib", "lab")
LABOBJ = os.path.join(ROOT, "worldforge_amd", "_build", "lab")

VARIANTS = {
    "base": [],                               # the shipped schedule
    # ablations -- results wrong by construction, they price one ingredient of the tile loop
    "ab_valu": ["-DWF_ATTN_ABLATE=1"],        # no softmax VALU
    "ab_bar": ["-DWF_ATTN_ABLATE=2"],         # no DMA drain / workgroup barrier
    "ab_lds": ["-DWF_ATTN_ABLATE=4"],         # no LDS fragment reads in the loop
    "nodma": ["-DWF_ATTN_DMA_PLACE=9"],       # no LDS-DMA pieces in the loop
    "ab_all": ["-DWF_ATTN_ABLATE=7", "-DWF_ATTN_DMA_PLACE=9"],   # MFMAs + the loop's scalar bookkeeping only
    # c_*: the same with light cycle counting (two s_memtime per workgroup): shader cycles per KV tile, independent of the clock the
    # power limit allows (removing work raises the clock, so TFLOP/s alone overstates what an ingredient costs)
    "c_base": ["-DWF_ATTN_TIMING=2"],
    "c_pf2": ["-DWF_ATTN_TIMING=2", "-DWF_ATTN_PF4=2"],   # round 4: fragment ring depth 2 / 8 instead of 4 (is the flight of the LDS reads a limit?)
    "c_pf8": ["-DWF_ATTN_TIMING=2", "-DWF_ATTN_PF4=8"],
    "c_valu": ["-DWF_ATTN_TIMING=2", "-DWF_ATTN_ABLATE=1"],
    "c_lds": ["-DWF_ATTN_TIMING=2", "-DWF_ATTN_ABLATE=4"],
    "c_dma": ["-DWF_ATTN_TIMING=2", "-DWF_ATTN_DMA_PLACE=9"],
    "c_bar": ["-DWF_ATTN_TIMING=2", "-DWF_ATTN_ABLATE=2"],
    "c_valu_lds": ["-DWF_ATTN_TIMING=2", "-DWF_ATTN_ABLATE=5"],
    "c_all": ["-DWF_ATTN_TIMING=2", "-DWF_ATTN_ABLATE=7", "-DWF_ATTN_DMA_PLACE=9"],
    "timing_base": ["-DWF_ATTN_TIMING"],      # s_memtime per 8-gap group (tools/attn_timing.py; heavy: the marks cost ~15 % themselves)
}


def build(names):
    from worldforge_amd import build as wb
    wb.build(verbose=True)                      # the normal library and every other object
    os.makedirs(LABLIB, exist_ok=True)
    os.makedirs(LABOBJ, exist_ok=True)
    cc = wb.hipcc()
    others = [os.path.join(wb.BUILD, s.replace(".hip", ".o")) for s in wb.SOURCES if s != "attention.hip"]

    def one(name):
        obj = os.path.join(LABOBJ, f"attention_{name}.o")
        cmd = [cc] + wb.COMMON + ["-DWF_ATTN_LAB"] + VARIANTS[name] + ["-c", os.path.join(wb.CSRC, "attention.hip"), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"{name}: {r.stderr[-3000:]}")
        lib = os.path.join(LABLIB, f"libwf_hip_{name}.so")
        r = subprocess.run([cc, "-shared", "-fPIC", f"--offload-arch={wb.ARCH}", "-o", lib, obj] + others + ["-L/opt/rocm/lib", "-lamdhip64"],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"{name}: link: {r.stderr[-3000:]}")
        print(f"[attn_lab] {lib}", flush=True)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(one, names))


def child(name, L, launches, outfile):
    import torch
    from worldforge_amd import dit
    H = 40
    Lp = (L + 63) // 64 * 64
    g = torch.Generator(device="cuda").manual_seed(7)
    q = torch.randn(H, L, 128, device="cuda", generator=g).bfloat16()
    k = torch.zeros(H, Lp, 128, device="cuda", dtype=torch.bfloat16)
    k[:, :L] = torch.randn(H, L, 128, device="cuda", generator=g).bfloat16()
    vt = torch.randn(H, Lp // 64, 128, 64, device="cuda", generator=g).bfloat16()
    out = torch.empty(L, H * 128, device="cuda", dtype=torch.bfloat16)
    qs = (q.float() * (1.4426950408889634 / math.sqrt(128))).bfloat16()
    km, qm = dit.head_max_norm2(k, L, torch.empty(H, device="cuda")), dit.head_max_norm2(qs, L, torch.empty(H, device="cuda"))
    import ctypes
    from worldforge_amd import _ffi
    lib = _ffi.lib()
    buf = (ctypes.c_ulonglong * 32)()
    res = {}
    for body, kk in (("untracked", km), ("tracked", None)):
        for _ in range(3):
            dit.attention(qs, k, vt, out, L, 0.0, kmax2=kk, qmax2=qm if kk is not None else None)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(launches):
            dit.attention(qs, k, vt, out, L, 0.0, kmax2=kk, qmax2=qm if kk is not None else None)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / launches
        res[body] = {"ms": ms, "tflops": 4.0 * L * L * 128 * H / ms / 1e9}
        if hasattr(lib, "wf_debug_attn_cycles"):      # a -DWF_ATTN_TIMING=2 build: shader cycles of the tile loop / tiles (wave 0 of every workgroup)
            lib.wf_debug_attn_cycles(buf, 1)
            dit.attention(qs, k, vt, out, L, 0.0, kmax2=kk, qmax2=qm if kk is not None else None)
            torch.cuda.synchronize()
            lib.wf_debug_attn_cycles(buf, 1)
            res[body]["cyc_per_tile"] = buf[20] / max(buf[19], 1)
        if outfile and body == "untracked":
            torch.save(out[::97].clone().cpu(), outfile)
    print("LAB " + json.dumps({"variant": name, **res}), flush=True)


def run(names, rounds, L, launches):
    import torch
    table = {n: [] for n in names}
    ref = None
    for r in range(rounds):
        for n in names:
            lib = os.path.join(LABLIB, f"libwf_hip_{n}.so")
            if not os.path.exists(lib):
                continue
            outfile = f"/tmp/attn_lab_{n}.pt" if r == 0 else ""
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "child", "--name", n, "--L", str(L), "--launches", str(launches),
                                "--outfile", outfile], env=dict(os.environ, WF_LIB=lib), capture_output=True, text=True, timeout=900)
            line = [l for l in p.stdout.splitlines() if l.startswith("LAB ")]
            if p.returncode != 0 or not line:
                print(f"{n}: FAILED rc={p.returncode}\n{p.stderr[-1500:]}", flush=True)
                continue
            d = json.loads(line[0][4:])
            table[n].append(d)
            diff = ""
            if r == 0:
                o = torch.load(outfile).float()
                if ref is None:
                    ref = o
                diff = f"  max|out - {names[0]}| = {float((o - ref).abs().max()):.3g}"
            cyc = ""
            if "cyc_per_tile" in d["untracked"]:
                cyc = f"  cycles/tile {d['untracked']['cyc_per_tile']:.0f} | {d['tracked']['cyc_per_tile']:.0f} (64 MFMAs = 2048)"
            print(f"round {r} {n:8s} untracked {d['untracked']['ms']:.3f} ms {d['untracked']['tflops']:.0f} TFLOP/s | tracked "
                  f"{d['tracked']['ms']:.3f} ms {d['tracked']['tflops']:.0f} TFLOP/s{cyc}{diff}", flush=True)
    print("\n| variant | un-tracked body TFLOP/s (rounds) | tracked body TFLOP/s (rounds) |\n|---|---|---|")
    for n in names:
        if table[n]:
            print(f"| {n} | " + " ".join(f"{d['untracked']['tflops']:.0f}" for d in table[n]) + " | "
                  + " ".join(f"{d['tracked']['tflops']:.0f}" for d in table[n]) + " |")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", choices=["build", "run", "child"])
    ap.add_argument("--variants", default=",".join(VARIANTS))
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--L", type=int, default=32760)
    ap.add_argument("--launches", type=int, default=20)
    ap.add_argument("--name", default="")
    ap.add_argument("--outfile", default="")
    a = ap.parse_args()
    names = [n for n in a.variants.split(",") if n]
    if a.mode == "build":
        build(names)
    elif a.mode == "run":
        run(names, a.rounds, a.L, a.launches)
    else:
        child(a.name, a.L, a.launches, a.outfile)
