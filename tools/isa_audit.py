"""Audit the generated gfx950 code of a kernel source for the two patterns that cost the most this round and that no profiler counter names:

  * conditional branches inside loops (or inside per-tile straight-line code) whose COMMON direction is "taken": a wave that is alone on its
    SIMD pays ~130 cycles of instruction fetch for each (attention: the ragged-tile test, 5 % of the loop; conv: ~96 per tile in the
    epilogue, three quarters of it) -- profiles/r3_q_attn_lab.md section 4;
  * loads that are waited for one at a time (`global_load` directly followed by `s_waitcnt vmcnt(0)`): a loop of load / modify / store
    over possibly-aliasing pointers serializes one memory round trip per iteration (GEMM residual epilogue: +6.7 % once batched).

    python tools/isa_audit.py worldforge_amd/csrc/conv.hip [--kernel k_conv_w4] [--flags=-DX]

No GPU needed (hipcc -S --cuda-device-only).  For every kernel: code bytes, registers / spills, and per loop (innermost first) the MFMA
count, conditional branches, loads, immediately-waited loads; then the same for the code outside loops.  It cannot know which direction of a
branch is the common one -- it lists them; read the blocks they skip.
"""
from __future__ import annotations

import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def disassemble(src: str, flags):
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    cmd = ["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-I", os.path.join(ROOT, "include"), "-S", "--cuda-device-only", "-o", out] + flags + [src]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(r.stderr[-3000:])
    text = open(out).read()
    os.unlink(out)
    return text


def kernels(text: str):
    """-> [(mangled name, [instruction lines])]"""
    res, name, body = [], None, []
    for line in text.split("\n"):
        m = re.match(r"^(_Z\w+):\s*; @", line)
        if m:
            name, body = m.group(1), []
            continue
        if name is not None:
            if line.startswith(".Lfunc_end"):
                res.append((name, body))
                name = None
            else:
                body.append(line)
    return res


def meta(text: str, name: str):
    m = re.search(r"\.amdhsa_kernel %s\b(.*?)\.end_amdhsa_kernel" % re.escape(name), text, re.S)
    d = {}
    if m:
        for key in ("next_free_vgpr", "accum_offset", "next_free_sgpr"):
            mm = re.search(r"\.amdhsa_%s (\d+)" % key, m.group(1))
            if mm:
                d[key] = int(mm.group(1))
    mm = re.search(r"; codeLenInByte = (\d+)", text[text.find(name + ":"):])
    if mm:
        d["code_bytes"] = int(mm.group(1))
    blk = text[text.find(name + ":"):]
    for key in ("ScratchSize", "VGPRSpill" if False else "ScratchSize"):
        mm = re.search(r"; %s: (\d+)" % key, blk)
        if mm:
            d[key] = int(mm.group(1))
    return d


def demangle(n: str) -> str:
    try:
        return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "") or n
    except OSError:
        return n


def analyse(body):
    """Loops from the compiler's own annotations (`; =>This Inner Loop Header` / `in Loop: Header=BBx_y`)."""
    label_of_line, loops = {}, {}
    cur = None
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", l)
        if m:
            cur = m.group(1)
            c = m.group(2) or ""
            hm = re.search(r"Header=(BB\d+_\d+)", c)
            if "Loop Header" in c:
                loops.setdefault(cur[2:], {"depth": int(re.search(r"Depth=(\d+)", c).group(1)), "lines": []})
                label_of_line[i] = cur[2:]
            elif hm:
                label_of_line[i] = hm.group(1)
            else:
                label_of_line[i] = None
    # assign every instruction to the loop of the most recent label
    region, where = None, []
    for i, l in enumerate(body):
        if i in label_of_line:
            region = label_of_line[i]
        where.append(region)
    stats = {}

    def bump(key, field, n=1):
        stats.setdefault(key, {"mfma": 0, "cbranch": 0, "branch": 0, "loads": 0, "waited_loads": 0, "barriers": 0, "insts": 0, "branch_lines": []})
        stats[key][field] += n

    prev_load = False
    for i, l in enumerate(body):
        t = l.strip()
        if not t or t.startswith((";", ".")) or re.match(r"^\.?\w+:", t):
            continue
        op = t.split()[0]
        key = where[i] or "(outside loops)"
        bump(key, "insts")
        if op.startswith("v_mfma"):
            bump(key, "mfma")
        if op.startswith("s_cbranch"):
            bump(key, "cbranch")
            stats[key]["branch_lines"].append((i, t))
        if op == "s_branch":
            bump(key, "branch")
        if op == "s_barrier":
            bump(key, "barriers")
        if re.match(r"(global|buffer|flat|scratch)_load", op) and "lds" not in op:
            bump(key, "loads")
            prev_load = True
            continue
        if prev_load and op == "s_waitcnt" and "vmcnt(0)" in t:
            bump(key, "waited_loads")
        prev_load = False
    return loops, stats


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("source")
    ap.add_argument("--kernel", default="", help="substring of the (demangled) kernel name")
    ap.add_argument("--flags", default="", help="extra hipcc flags, space separated")
    ap.add_argument("--branches", action="store_true", help="list every conditional branch of the selected kernels")
    a = ap.parse_args()
    text = disassemble(a.source, a.flags.split())
    for name, body in kernels(text):
        pretty = demangle(name)
        if a.kernel and a.kernel not in pretty:
            continue
        md = meta(text, name)
        print(f"== {pretty}\n   code {md.get('code_bytes', '?')} B, vgpr {md.get('next_free_vgpr', '?')} (agpr from {md.get('accum_offset', '-')}), "
              f"scratch {md.get('ScratchSize', '?')} B")
        loops, stats = analyse(body)
        order = sorted(stats, key=lambda k: (-(loops.get(k, {}).get("depth", 0)), k))
        for k in order:
            s = stats[k]
            d = loops.get(k, {}).get("depth")
            tag = f"loop {k} depth {d}" if d else k
            flag = ""
            if s["mfma"] and s["cbranch"] > 2:
                flag += "  <- conditional branches in an MFMA loop: make sure the common case falls through"
            if s["waited_loads"] >= 4:
                flag += "  <- loads waited for one at a time"
            print(f"   {tag:28s} insts {s['insts']:6d}  mfma {s['mfma']:4d}  cond.branches {s['cbranch']:4d}  branches {s['branch']:3d}  barriers {s['barriers']:2d}  "
                  f"loads {s['loads']:4d} (waited singly {s['waited_loads']}){flag}")
            if a.branches:
                for i, t in s["branch_lines"]:
                    print(f"        line {i}: {t}")


if __name__ == "__main__":
    sys.exit(main())
