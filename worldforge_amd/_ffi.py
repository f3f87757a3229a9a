"""ctypes binding of libwf_hip.so.

The prototypes are parsed from include/wf_hip.h so that the Python side can never drift from the C-ABI.  There is no
CPU fallback: if the shared library is missing or a call fails, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes
import os
import re
from typing import Dict, List, Tuple

# Import torch BEFORE dlopen()ing libwf_hip.so: PyTorch-ROCm ships its own libamdhip64.so.7 / libhsa-runtime64 (rocm 7.0)
# under torch/lib, and /opt/rocm carries 7.2 with the same SONAME.  Whichever is loaded first serves both; if ours pulled in
# /opt/rocm's copy first, torch would run on a runtime / HSA pair it was not built with ("no ROCm-capable device").
import torch  # noqa: F401

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
HEADER = os.path.join(ROOT, "include", "wf_hip.h")
LIB_PATH = os.environ.get("WF_LIB") or os.path.join(HERE, "_lib", "libwf_hip.so")  # WF_LIB: an instrumented build (tools/*_timing.py)

WF_F32, WF_BF16 = 0, 1

_CTYPES = {
    "int": ctypes.c_int,
    "float": ctypes.c_float,
    "double": ctypes.c_double,
    "size_t": ctypes.c_size_t,
    "int64_t": ctypes.c_int64,
    "uint64_t": ctypes.c_uint64,
}


def parse_header(path: str = HEADER) -> Dict[str, Tuple[object, List[object]]]:
    """Return {symbol: (restype, [argtypes])} for every `wf_*` prototype in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    src = re.sub(r"^\s*#[^\n]*", " ", src, flags=re.M)  # preprocessor lines
    src = src.replace('extern "C" {', " ")
    protos: Dict[str, Tuple[object, List[object]]] = {}
    for m in re.finditer(r"([A-Za-z_][\w \t\*]*?)\b(wf_\w+)\s*\(([^)]*)\)\s*;", src):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if "*" in ret:
            restype = ctypes.c_char_p if "char" in ret else ctypes.c_void_p
        else:
            restype = _CTYPES[ret.replace("const", "").strip()]
        argtypes: List[object] = []
        if args and args != "void":
            for a in args.split(","):
                a = a.strip()
                if "*" in a:
                    argtypes.append(ctypes.c_char_p if re.search(r"\bchar\b", a) else ctypes.c_void_p)
                else:
                    ty = re.sub(r"\bconst\b", "", a).split()
                    argtypes.append(_CTYPES[ty[0]])
        protos[name] = (restype, argtypes)
    return protos


class _Lib:
    def __init__(self):
        self._dll = None
        self._protos = None

    def load(self):
        if self._dll is not None:
            return self._dll
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"libwf_hip.so not found at {LIB_PATH}; run `python -m worldforge_amd.build` (hipcc, gfx950). "
                "There is no CPU fallback for the hot path."
            )
        dll = ctypes.CDLL(LIB_PATH)
        self._protos = parse_header()
        for name, (restype, argtypes) in self._protos.items():
            fn = getattr(dll, name)  # AttributeError -> the .so is stale vs the header
            fn.restype = restype
            fn.argtypes = argtypes
        self._dll = dll
        return dll

    @property
    def protos(self):
        if self._protos is None:
            self._protos = parse_header()
        return self._protos


_LIB = _Lib()


def lib():
    return _LIB.load()


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().wf_last_error().decode(errors="replace")
        raise RuntimeError(f"libwf_hip {what} failed (rc={rc}): {msg}")


def call(name: str, *args):
    """Call an int-returning entry point and raise on a non-zero status."""
    fn = getattr(lib(), name)
    rc = fn(*args)
    if rc != 0:
        check(rc, name)


def farr(vals):
    """Host float array for the few by-pointer host arguments (means/stds)."""
    return (ctypes.c_float * len(vals))(*[float(v) for v in vals])


def iarr(vals):
    return (ctypes.c_int * len(vals))(*[int(v) for v in vals])
