"""Register / scratch usage of every kernel in a built object of worldforge_amd/_build (no GPU needed).

    python tools/kernel_resources.py conv attention      # objects by stem; no argument = all
    python tools/kernel_resources.py conv --isa k_conv_w4   # + count of scratch_* / v_readlane / v_writelane / s_waitcnt in matching kernels

Reads the AMDGPU metadata notes (llvm-readelf --notes) of the gfx950 code object inside the .o's .hip_fatbin section.
"""
from __future__ import annotations

import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "worldforge_amd", "_build")
LLVM = "/opt/rocm/lib/llvm/bin"


def code_object(obj: str, tmp: str) -> str:
    fat, co = os.path.join(tmp, "fat.bin"), os.path.join(tmp, os.path.basename(obj) + ".co")
    subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", obj], check=True)
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
    return co


def notes(co: str):
    txt = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    kernels = []
    for blk in re.split(r"\n\s*- \.agpr_count:", txt)[1:]:
        blk = ".agpr_count:" + blk
        get = lambda k: (re.search(rf"\.{k}:\s*(\S+)", blk) or [None, "?"])[1]  # noqa: E731
        kernels.append({k: get(k) for k in ("name", "vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count",
                                            "private_segment_fixed_size", "group_segment_fixed_size")})
    return kernels


def demangle(n: str) -> str:
    try:
        return subprocess.run([f"{LLVM}/llvm-cxxfilt", n], capture_output=True, text=True).stdout.strip() or n
    except OSError:
        return n


def isa_counts(co: str, pattern: str):
    dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--mcpu=gfx950", co], check=True, capture_output=True, text=True).stdout
    out, cur = {}, None
    for line in dis.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
        if m:
            cur = m.group(1) if pattern in demangle(m.group(1)) else None
            if cur:
                out[cur] = {"scratch": 0, "v_readlane": 0, "v_writelane": 0, "s_waitcnt": 0, "v_mfma": 0, "insts": 0}
            continue
        if cur and "\t" in line:
            out[cur]["insts"] += 1
            for k in ("scratch", "v_readlane", "v_writelane", "s_waitcnt", "v_mfma"):
                if re.search(rf"\b{k}", line):
                    out[cur][k] += 1
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("stems", nargs="*")
    ap.add_argument("--isa", default=None, help="substring of the demangled kernel name: count spill / wait instructions in its ISA")
    a = ap.parse_args()
    stems = a.stems or sorted(f[:-2] for f in os.listdir(BUILD) if f.endswith(".o"))
    with tempfile.TemporaryDirectory() as tmp:
        for s in stems:
            co = code_object(os.path.join(BUILD, s + ".o"), tmp)
            print(f"== {s}.o")
            for k in notes(co):
                print(f"  {demangle(k['name'])[:110]:110s} vgpr {k['vgpr_count']:>4} agpr {k['agpr_count']:>4} sgpr {k['sgpr_count']:>4} "
                      f"vspill {k['vgpr_spill_count']:>3} sspill {k['sgpr_spill_count']:>3} scratch {k['private_segment_fixed_size']:>4} B "
                      f"lds {k['group_segment_fixed_size']}")
            if a.isa:
                for n, c in isa_counts(co, a.isa).items():
                    print(f"  ISA {demangle(n)[:100]}: {c}")
    sys.exit(0)
