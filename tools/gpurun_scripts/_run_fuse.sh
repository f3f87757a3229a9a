#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_dit.py -q -x 2>&1 | grep -E "passed|failed" > gpurun_out/fuse_ab.txt
python - >> gpurun_out/fuse_ab.txt 2>&1 <<'PY'
import torch
from worldforge_amd import dit
for L in (4096, 32760):
    Lp = (L + 63) // 64 * 64
    k = torch.randn(40, Lp, 128, device="cuda").bfloat16()
    out = torch.empty(40, device="cuda")
    for _ in range(3): dit.head_max_norm2(k, L, out)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): dit.head_max_norm2(k, L, out)
    e.record(); torch.cuda.synchronize()
    ref = (k[:, :L].float() ** 2).sum(-1).max(-1).values
    print(f"L={L}: {s.elapsed_time(e) / 20 * 1e3:.1f} us per call (incl. the zero fill), {40 * L * 256 / (s.elapsed_time(e) / 20 * 1e-3) / 1e12:.2f} TB/s, equal={torch.equal(out, ref) or float((out - ref).abs().max())}")
PY
grep -v amdgpu gpurun_out/fuse_ab.txt
