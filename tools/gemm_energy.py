"""GEMM energy / clock table (VERDICT r2 "Next" #5): TFLOP/s, average socket power and shader clock (rocm-smi, sampled while the work runs)
for (a) the MFMA lab streams of tools/gemm_lab/mfma_lab.hip -- matrix pipe only, registers or LDS operands, zeros vs N(0,1) -- and (b) the
library's GEMM kernels on the DiT's QKV shape (M = 32760, N = 15360, K = 5120): ping-pong 256- / 320-wide tiles, the one-wave-per-SIMD
kernel, zeros vs N(0,1) operands.      python tools/gemm_energy.py > gpurun_out/.../gemm_energy.md
"""
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LAB = os.path.join(ROOT, "tools", "gemm_lab", "mfma_lab")

GEMM_CHILD = r"""
import os, sys, time, torch
sys.path.insert(0, %r)
from worldforge_amd import dit
M, N, K = 32760, 15360, 5120
zeros = os.environ.get("DATA") == "zeros"
x = (torch.zeros if zeros else torch.randn)(M, K, device="cuda").bfloat16()
w = ((torch.zeros if zeros else torch.randn)(N, K, device="cuda") / K ** 0.5).bfloat16()
b = torch.zeros(N, device="cuda")
out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
for _ in range(3): dit.gemm(x, w, b, out, 0)
torch.cuda.synchronize()
print("READY", flush=True)
t_end = time.time() + float(os.environ.get("SECONDS", "4"))
n, ms = 0, 0.0
while time.time() < t_end:
    a, c = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): dit.gemm(x, w, b, out, 0)
    c.record(); torch.cuda.synchronize()
    ms += a.elapsed_time(c); n += 20
print("RESULT %%.0f TFLOP/s, %%.3f ms per launch" %% (2.0 * M * N * K * n / ms / 1e9, ms / n), flush=True)
""" % ROOT


def smi():
    """(power W, sclk MHz) from rocm-smi, or (None, None)."""
    try:
        r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=20)
        d = json.loads(r.stdout)
        card = next(iter(d.values()))
        p = next((float(v) for k, v in card.items() if "ower" in k and re.match(r"^[0-9.]+$", str(v))), None)
        s = next((v for k, v in card.items() if "sclk" in k.lower()), None)
        m = re.search(r"(\d+)\s*Mhz", str(s), re.I)
        return p, (int(m.group(1)) if m else None)
    except Exception:
        return None, None


def run(label, cmd, env=None, seconds=4.0):
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if cmd[0] == sys.executable:   # python children: wait until the timed loop has started (they print READY), then sample inside it
        while True:
            line0 = p.stdout.readline()
            if not line0 or line0.startswith("READY"):
                break
        time.sleep(0.4 * seconds)
    else:
        time.sleep(0.55 * seconds)
    samples = [smi() for _ in range(3)]
    out = p.communicate()[0]
    line = next((l for l in out.splitlines() if l.startswith("RESULT") or l.startswith("variant")), out.strip().splitlines()[-1] if out.strip() else "")
    pw = [s[0] for s in samples if s[0] is not None]
    ck = [s[1] for s in samples if s[1] is not None]
    print(f"| {label} | {line.replace('RESULT ', '')} | {sum(pw) / len(pw):.0f} W | {sum(ck) / len(ck):.0f} MHz |" if pw and ck else
          f"| {label} | {line.replace('RESULT ', '')} | n/a | n/a |", flush=True)


if __name__ == "__main__":
    print("| workload | result | socket power (rocm-smi, mean of 3 samples mid-run) | sclk (rocm-smi) |\n|---|---|---|---|", flush=True)
    p, c = smi()
    print(f"| idle | - | {p} W | {c} MHz |", flush=True)
    names = {0: "lab: 32x32x16, register operands", 1: "lab: 16x16x32, register operands", 2: "lab: 32x32x16 + 0.5 ds_read_b128 per MFMA",
             3: "lab: 32x32x16 + 0.75 ds_read_b128 per MFMA", 4: "lab: 32x32x16, two waves per SIMD, register operands"}
    for var in (0, 1, 2, 3, 4):
        for data in (0, 1):
            run(f"{names[var]}, {'N(0,1)' if data else 'zeros'}", [LAB, str(var), str(data), "4"])
    # (round 5: `k_gemm_w4` / `k_gemm_pp16` were removed from the library; WF_GEMM_TILE is read by lab builds only, -DWF_GEMM_LAB_TILE)
    for kern, env_k in (("ping-pong 320-wide tile (default)", {}), ("ping-pong 256-wide tile (lab build)", {"WF_GEMM_TILE": "256"})):
        for data in ("zeros", "random"):
            env = dict(os.environ, DATA=data, SECONDS="6", **env_k)
            run(f"QKV GEMM 32760 x 15360 x 5120, {kern}, {'N(0,1)' if data == 'random' else 'zeros'}", [sys.executable, "-c", GEMM_CHILD], env=env, seconds=6.0)
