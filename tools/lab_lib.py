"""Build a lab library worldforge_amd/_lib/lab/libwf_hip_<name>.so that is the current library with some sources taken from a git revision:

    python tools/lab_lib.py conv_old conv.hip=HEAD            # conv.hip as committed, everything else from the working tree
    python tools/lab_lib.py both_old conv.hip=HEAD gemm.hip=HEAD~1 mfma.h=HEAD

Headers named this way replace the working-tree header for the swapped SOURCES only.  The other objects are the ones `python -m worldforge_amd.build`
left in worldforge_amd/_build (run it first).  Use with WF_LIB=<path> for same-box A/B runs (tools/gpurun_scripts)."""
import os, subprocess, sys, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from worldforge_amd import build as B


def main():
    name, swaps = sys.argv[1], dict(a.split("=") for a in sys.argv[2:])
    extra = os.environ.get("WF_EXTRA_HIPCC_FLAGS", "").split()  # instrumentation flags apply to the swapped sources ONLY: the shared
    for f in extra:                                               # objects (and libwf_hip.so) stay the plain build
        B.COMMON.remove(f)
    B.build(verbose=False)
    cc = B.hipcc()
    out_dir = os.path.join(B.LIBDIR, "lab")
    os.makedirs(out_dir, exist_ok=True)
    top = tempfile.mkdtemp(prefix="wf_lab_")
    tmp = os.path.join(top, "worldforge_amd", "csrc")   # common.h includes "../../include/wf_hip.h"
    os.makedirs(tmp)
    shutil.copytree(os.path.join(B.ROOT, "include"), os.path.join(top, "include"))
    try:
        for f in os.listdir(B.CSRC):  # a private copy of csrc with the swapped files
            shutil.copy(os.path.join(B.CSRC, f), os.path.join(tmp, f))
        for f, rev in swaps.items():
            if rev == "WORK":  # the working-tree file, recompiled (with WF_EXTRA_HIPCC_FLAGS, e.g. -DWF_CONV_TIMING -DWF_CONV_ABLATE)
                continue
            src = subprocess.run(["git", "show", f"{rev}:worldforge_amd/csrc/{f}"], cwd=B.ROOT, capture_output=True, check=True).stdout
            open(os.path.join(tmp, f), "wb").write(src)
        hdr_swapped = any(f.endswith(".h") for f in swaps)
        objs = []
        for s, flags in B.SOURCES.items():
            if s in swaps or (hdr_swapped and s in ("conv.hip", "gemm.hip", "attention.hip")):
                obj = os.path.join(tmp, s.replace(".hip", ".o"))
                subprocess.run([cc] + B.COMMON + extra + flags + ["-c", os.path.join(tmp, s), "-o", obj], check=True)
            else:
                obj = os.path.join(B.BUILD, s.replace(".hip", ".o"))
            objs.append(obj)
        lib = os.path.join(out_dir, f"libwf_hip_{name}.so")
        subprocess.run([cc, "-shared", "-fPIC", f"--offload-arch={B.ARCH}", "-o", lib] + objs + ["-L/opt/rocm/lib", "-lamdhip64"], check=True)
        print(lib)
    finally:
        shutil.rmtree(top, ignore_errors=True)


if __name__ == "__main__":
    main()
