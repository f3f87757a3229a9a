"""CPU: the host halves of every C-ABI entry point under AddressSanitizer + UndefinedBehaviorSanitizer (the GPU
sanitizers are unavailable on the pool).  tools/sanitize_host.py compiles csrc/*.hip host-only with the sanitizers, links a stub HIP
runtime whose hipMemsetAsync writes the HOST buffer it is given, proves the harness catches an undersized workspace, then drives all
entry points with null / bad / exactly-sized arguments."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_host_side_of_the_abi_is_clean_under_asan_and_ubsan():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "sanitize_host.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "self-test ok" in r.stdout and "sanitize_host: clean" in r.stdout
    assert " 0 bad launch geometries" in r.stdout
