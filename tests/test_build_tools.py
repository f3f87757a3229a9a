"""Host-side consistency of the measurement tooling with the kernel sources (no GPU, no compile)."""
import importlib.util
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_attn_lab_variants_name_macros_the_kernel_source_knows():
    """tools/attn_lab.py builds one library per -D variant of attention.hip: a flag the source no longer reads would silently build the
    shipped kernel under a variant's name and the lab would report "no difference"."""
    lab = _load(os.path.join(ROOT, "tools", "attn_lab.py"), "attn_lab")
    src = open(os.path.join(ROOT, "worldforge_amd", "csrc", "attention.hip")).read()
    assert "base" in lab.VARIANTS and lab.VARIANTS["base"] == []
    for name, flags in lab.VARIANTS.items():
        for f in flags:
            m = re.fullmatch(r"-D(WF_ATTN_\w+)(=\w+)?", f)
            assert m, (name, f)
            assert re.search(r"\b%s\b" % m.group(1), src), f"{name}: attention.hip does not read {m.group(1)}"
    assert "WF_ATTN_LAB" in src  # the switch that restricts a lab build to the timed kernel


def test_gpurun_scripts_readme_lists_every_script():
    d = os.path.join(ROOT, "tools", "gpurun_scripts")
    readme = open(os.path.join(d, "README.md")).read()
    ranges = [(r, a, b) for r, a, b in re.findall(r"`r(\d)_(\w)\.sh` \.\.\. `r\1_(\w)\.sh`", readme)]
    for f in sorted(os.listdir(d)):
        if not f.endswith(".sh"):
            continue
        m = re.fullmatch(r"r(\d)_(\w)\.sh", f)
        in_range = bool(m) and any(r == m.group(1) and a <= m.group(2) <= b for r, a, b in ranges)
        assert f"`{f}`" in readme or in_range, f"{f} is not described in tools/gpurun_scripts/README.md"
