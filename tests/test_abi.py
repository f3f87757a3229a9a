"""CPU: the C-ABI library builds, loads and exports every symbol include/wf_hip.h declares (no compute calls)."""
import ctypes
import os

from worldforge_amd import _ffi, build


def test_library_builds_and_exports_every_declared_symbol():
    path = build.build(verbose=False)
    assert os.path.exists(path)
    dll = ctypes.CDLL(path)
    protos = _ffi.parse_header()
    assert len(protos) >= 19
    missing = [n for n in protos if not hasattr(dll, n)]
    assert not missing, missing
    assert _ffi.lib().wf_version() >= 100


def test_host_only_entry_points():
    lib = _ffi.lib()
    assert lib.wf_dsg_workspace_floats() >= 8
    assert lib.wf_flow_metrics_workspace_floats(16) >= 16 * 3
    # argument validation happens before any device work
    assert lib.wf_cfg_combine(None, None, None, 0, 1.0, 4, None) != 0
    assert b"null" in lib.wf_last_error()


def test_vae_calls_never_synchronise_the_host():
    """no hidden device syncs.  Since round 5 AutoencoderKLWan.encode / decode / decode_blend_encode end with an ASYNCHRONOUS
    copy of the fp16 range flag (wf_f16_overflow_flag_async + an event) instead of the synchronous read of round 4: neither those methods
    nor _note_range may name the synchronous entry point or synchronise anything, and the asynchronous entry point's C body must not
    contain a stream / device synchronisation.  (The behaviour itself -- decode returns while the GPU is still busy -- is asserted on
    the GPU in tests/test_gpu_vae.py.)"""
    import ast
    import re
    src = open(os.path.join(os.path.dirname(_ffi.HERE), "worldforge_amd", "vae.py")).read()
    tree = ast.parse(src)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "AutoencoderKLWan")
    fns = {n.name: ast.get_source_segment(src, n) for n in cls.body if isinstance(n, ast.FunctionDef)}
    for name in ("encode", "decode", "decode_blend_encode", "_note_range", "_encode_one", "_decode_one"):
        body = fns[name]
        assert '"wf_f16_overflow_flag"' not in body and "synchronize" not in body and ".item()" not in body and ".cpu()" not in body, name
    assert "wf_f16_overflow_flag_async" in fns["_note_range"]
    csrc = open(os.path.join(os.path.dirname(_ffi.HERE), "worldforge_amd", "csrc", "vae_ops.hip")).read()
    m = re.search(r'extern "C" int wf_f16_overflow_flag_async\(.*?\n}\n', csrc, re.S)
    assert m and "Synchronize" not in m.group(0) and "hipMemcpyFromSymbolAsync" in m.group(0)
