"""CPU sanitizer pass over the HOST halves of libwf_hip.so (GPU AddressSanitizer is not available on the pool).

    python tools/sanitize_host.py            # build + run, prints a summary, exit code 0 = clean

What it does
  1. compiles every csrc/*.hip with `hipcc --offload-host-only -fsanitize=address,undefined` (host code only: argument validation, shape
     arithmetic, workspace carving, launch-geometry computation, the per-call argument structs; device code is not compiled in);
  2. links the objects against a STUB HIP runtime generated here (tools/_sanitize/hip_stub.c): `hipLaunchKernel` records the launch
     geometry and returns success without running anything, `hipMemsetAsync` performs the memset ON THE HOST BUFFER IT IS GIVEN -- so a
     workspace carved past the size `wf_*_workspace_bytes` reported is a heap-buffer-overflow ASan reports -- and the fat-binary
     registration hooks are no-ops;
  3. runs tools/_sanitize driver in a child python with the ASan runtime pre-loaded: every entry point of include/wf_hip.h is called
     (a) with null / mis-sized / mis-aligned arguments (must return an error code and set wf_last_error, never touch memory) and
     (b) with valid host buffers sized exactly by the query functions (must return 0; every launch must have a non-empty grid).
Any ASan / UBSan report makes the child exit non-zero.  This checks the host side only; the kernels are covered by the GPU parity tests.
"""
from __future__ import annotations

import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "worldforge_amd", "csrc")
OUT = os.path.join(ROOT, "worldforge_amd", "_build", "asan")
LIB = os.path.join(OUT, "libwf_hip_asan.so")
FP_OFF = {"elementwise", "inject", "flow", "longcat_ops", "warp", "crackfill", "pointrender"}

STUB = r"""
#include <string.h>
#include <stddef.h>
#include <stdint.h>
typedef struct { unsigned x, y, z; } dim3_;
static dim3_ g_grid, g_block; static size_t g_shmem; static void* g_stream;
unsigned long long wf_stub_launches = 0, wf_stub_empty_grids = 0, wf_stub_max_threads = 0;
void** __hipRegisterFatBinary(const void* d) { static void* h; (void)d; return &h; }
void __hipUnregisterFatBinary(void** h) { (void)h; }
void __hipRegisterFunction(void** m, const void* f, char* a, const char* b, unsigned c, void* d, void* e, void* g, void* i, int* w) {
  (void)m; (void)f; (void)a; (void)b; (void)c; (void)d; (void)e; (void)g; (void)i; (void)w; }
void __hipRegisterVar(void** m, void* v, char* a, const char* b, int e, size_t s, int c, int g) { (void)m; (void)v; (void)a; (void)b; (void)e; (void)s; (void)c; (void)g; }
int __hipPushCallConfiguration(dim3_ grid, dim3_ block, size_t shmem, void* stream) { g_grid = grid; g_block = block; g_shmem = shmem; g_stream = stream; return 0; }
int __hipPopCallConfiguration(dim3_* grid, dim3_* block, size_t* shmem, void** stream) { *grid = g_grid; *block = g_block; *shmem = g_shmem; *stream = g_stream; return 0; }
int hipLaunchKernel(const void* f, dim3_ grid, dim3_ block, void** args, size_t shmem, void* stream) {
  (void)f; (void)args; (void)stream;
  ++wf_stub_launches;
  if (grid.x == 0 || grid.y == 0 || grid.z == 0 || block.x == 0) ++wf_stub_empty_grids;
  unsigned long long t = (unsigned long long)block.x * block.y * block.z;
  if (t > wf_stub_max_threads) wf_stub_max_threads = t;
  if (shmem > 160u * 1024u) ++wf_stub_empty_grids;   /* more LDS than a CU has: also a host bug */
  return 0; }
int hipGetLastError(void) { return 0; }
const char* hipGetErrorString(int e) { (void)e; return "stub"; }
int hipMemsetAsync(void* p, int v, size_t n, void* s) { (void)s; memset(p, v, n); return 0; }
int hipGetDevice(int* d) { *d = 0; return 0; }
int hipDeviceGetAttribute(int* v, int a, int d) { (void)a; (void)d; *v = 256; return 0; }
int hipGetDevicePropertiesR0600(void* p, int d) { (void)p; (void)d; return 100; }   /* "no device": wf_device_info must report the error */
int hipMemcpyToSymbol(const void* s, const void* src, size_t n, size_t o, int k) { (void)s; (void)src; (void)n; (void)o; (void)k; return 0; }
int hipMemcpyFromSymbol(void* d, const void* s, size_t n, size_t o, int k) { (void)s; (void)o; (void)k; memset(d, 0, n); return 0; }
int hipMemcpyFromSymbolAsync(void* d, const void* s, size_t n, size_t o, int k, void* st) { (void)s; (void)o; (void)k; (void)st; memset(d, 0, n); return 0; }
"""


def sh(cmd, **kw):
    r = subprocess.run(cmd, capture_output=True, text=True, **kw)
    if r.returncode != 0:
        raise RuntimeError(f"{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
    return r.stdout


def build():
    os.makedirs(OUT, exist_ok=True)
    objs = []
    for f in sorted(os.listdir(CSRC)):
        if not f.endswith(".hip"):
            continue
        stem = f[:-4]
        obj = os.path.join(OUT, stem + ".o")
        flags = ["-ffp-contract=off"] if stem in FP_OFF else []
        sh(["hipcc", "-O1", "-g", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "--offload-host-only", "-fsanitize=address,undefined",
            "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-Wno-unused-result", "-I", os.path.join(ROOT, "include")] + flags
           + ["-c", os.path.join(CSRC, f), "-o", obj])
        objs.append(obj)
    undef = sh(["nm", "-u"] + objs)
    fatbins = sorted(set(re.findall(r"U (__hip_fatbin_\w+)", undef)))
    stub_c = os.path.join(OUT, "hip_stub.c")
    with open(stub_c, "w") as fh:
        fh.write(STUB + "\n" + "\n".join(f"const char {s}[64] = {{0}};" for s in fatbins) + "\n")
    sh(["gcc", "-O1", "-g", "-fPIC", "-c", stub_c, "-o", os.path.join(OUT, "hip_stub.o")])
    sh(["hipcc", "-shared", "-fPIC", "-fsanitize=address,undefined", "-o", LIB] + objs + [os.path.join(OUT, "hip_stub.o")])
    return LIB


def asan_runtime():
    out = sh(["hipcc", "-print-file-name=libclang_rt.asan-x86_64.so"]).strip()
    if os.path.isabs(out) and os.path.exists(out):
        return out
    for base, _, files in os.walk("/opt/rocm/lib/llvm/lib/clang"):
        for f in files:
            if f == "libclang_rt.asan-x86_64.so":
                return os.path.join(base, f)
    raise RuntimeError("ASan runtime not found")


def main():
    lib = build()
    env = dict(os.environ, LD_PRELOAD=asan_runtime(), WF_LIB=lib, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=66",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    # self-test first: an undersized workspace must be reported (exit code 66 = ASan), otherwise a clean run below would mean nothing
    st = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "_sanitize_driver.py")], env=dict(env, WF_SANITIZE_SELFTEST="1"),
                        capture_output=True, text=True)
    if st.returncode != 66 or "heap-buffer-overflow" not in st.stderr:
        sys.stderr.write(st.stdout[-2000:] + st.stderr[-4000:])
        print("sanitize_host: SELF-TEST FAILED -- the harness does not detect an undersized workspace")
        return 2
    print("sanitize_host: self-test ok (an undersized wf_crack_fill workspace is reported as heap-buffer-overflow)")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "_sanitize_driver.py")], env=env, capture_output=True, text=True)
    sys.stdout.write(r.stdout[-6000:])
    if r.returncode != 0:
        sys.stderr.write(r.stderr[-8000:])
        print(f"sanitize_host: FAILED (exit {r.returncode})")
        return 1
    print("sanitize_host: clean")
    return 0


if __name__ == "__main__":
    sys.exit(main())
