"""Build libwf_hip.so (the C-ABI in include/wf_hip.h) for gfx950 with hipcc.

In-tree build: objects go to worldforge_amd/_build/, the shared library to worldforge_amd/_lib/libwf_hip.so
(git-ignored, but shipped to the GPU box by gpurun).  hipcc cross-compiles gfx950 without a GPU.
Usage:  python -m worldforge_amd.build [--force] [--jobs N]
"""
from __future__ import annotations

import argparse
import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
BUILD = os.path.join(HERE, "_build")
LIBDIR = os.path.join(HERE, "_lib")
LIB = os.path.join(LIBDIR, "libwf_hip.so")

ARCH = "gfx950"
COMMON = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wno-unused-result", "-I", os.path.join(ROOT, "include")]
COMMON += os.environ.get("WF_EXTRA_HIPCC_FLAGS", "").split()  # debug builds (e.g. -DWF_ATTN_TIMING)
# Per-file extra flags.  The element-wise / injection kernels must round every op separately (no FMA contraction) to
# reproduce eager PyTorch bit for bit.
SOURCES = {
    "elementwise.hip": ["-ffp-contract=off"],
    "inject.hip": ["-ffp-contract=off"],
    "flow.hip": ["-ffp-contract=off"],
    "gemm.hip": [],
    "attention.hip": [],
    "dit_ops.hip": [],
    "longcat_ops.hip": ["-ffp-contract=off"],
    "vae_ops.hip": [],
    "conv.hip": [],
    "warp.hip": ["-ffp-contract=off"],
    "crackfill.hip": ["-ffp-contract=off"],
    "pointrender.hip": ["-ffp-contract=off"],
    "calib.hip": [],
}


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


def _digest(paths, flags) -> str:
    h = hashlib.sha256()
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(flags).encode())
    return h.hexdigest()


def _headers():
    hs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h")]
    hs.append(os.path.join(ROOT, "include", "wf_hip.h"))
    return hs


def _compile_one(cc, src, flags, force):
    path = os.path.join(CSRC, src)
    obj = os.path.join(BUILD, src.replace(".hip", ".o"))
    stamp = obj + ".sha"
    dig = _digest([path] + _headers(), COMMON + flags)
    if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig:
        return obj, False
    cmd = [cc] + COMMON + flags + ["-c", path, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {src}:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
    with open(stamp, "w") as f:
        f.write(dig)
    return obj, True


def build(force: bool = False, jobs: int = 4, verbose: bool = True) -> str:
    cc = hipcc()
    os.makedirs(BUILD, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    present = {s: f for s, f in SOURCES.items() if os.path.exists(os.path.join(CSRC, s))}
    with ThreadPoolExecutor(max_workers=max(1, jobs)) as ex:
        results = list(ex.map(lambda kv: _compile_one(cc, kv[0], kv[1], force), present.items()))
    objs = [o for o, _ in results]
    changed = any(c for _, c in results)
    if changed or force or not os.path.exists(LIB):
        libs = ["-L/opt/rocm/lib", "-lamdhip64"]
        if "comm.hip" in present:
            libs += ["-lrccl"]
        cmd = [cc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs + libs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
        if verbose:
            print(f"[wf build] linked {LIB} ({len(objs)} objects)")
    elif verbose:
        print(f"[wf build] up to date: {LIB}")
    return LIB


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--force", action="store_true")
    ap.add_argument("--jobs", type=int, default=4)
    a = ap.parse_args()
    try:
        build(force=a.force, jobs=a.jobs)
    except RuntimeError as e:
        print(e, file=sys.stderr)
        sys.exit(1)
