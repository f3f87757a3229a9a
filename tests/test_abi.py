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
