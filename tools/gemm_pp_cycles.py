"""Cycle shares of the ping-pong GEMM `k_gemm_pp` per output tile: prologue | K loop | epilogue (what would hiding
the epilogue buy?).  Shader cycles of wave 0 of every workgroup (`-DWF_GEMM_TIMING` marks in gemm.hip), next to the launch's wall time.

    python tools/gemm_pp_cycles.py build     # here: worldforge_amd/_lib/lab/libwf_hip_gemmtiming.so (gemm.hip with -DWF_GEMM_TIMING + the normal objects)
    python tools/gemm_pp_cycles.py run       # on the GPU box (child process with WF_LIB = that library)
"""
import ctypes, math, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LABDIR = os.path.join(ROOT, "worldforge_amd", "_lib", "lab")
LIB = os.path.join(LABDIR, "libwf_hip_gemmtiming.so")
# ablations of the K loop (wrong results by construction: they price one ingredient in cycles, like tools/attn_lab.py does for the attention)
VARIANTS = {"gemmtiming": [], "gemm_touch1": ["-DWF_GEMM_TOUCH=1"], "gemm_touch2": ["-DWF_GEMM_TOUCH=2"], "gemm_touch2_24": ["-DWF_GEMM_TOUCH=2", "-DWF_GEMM_TOUCH_AT=24"], "gemm_dma_in_mfma": ["-DWF_GEMM_DMA_PHASE=0"], "gemm_dma_in_read": ["-DWF_GEMM_DMA_PHASE=1"], "gemm_rsplit0": ["-DWF_GEMM_DMA_RSPLIT=0"], "gemm_rsplit3": ["-DWF_GEMM_DMA_RSPLIT=3"], "gemm_rsplit7": ["-DWF_GEMM_DMA_RSPLIT=7"], "gemm_rsplit9": ["-DWF_GEMM_DMA_RSPLIT=9"], "gemm_nodma": ["-DWF_GEMM_ABLATE=1"], "gemm_nophasebar": ["-DWF_GEMM_ABLATE=2"], "gemm_nolds": ["-DWF_GEMM_ABLATE=4"],
            "gemm_noprio": ["-DWF_GEMM_ABLATE=8"], "gemm_mfma_only": ["-DWF_GEMM_ABLATE=7"]}


def build():
    from worldforge_amd import build as wb
    wb.build(verbose=True)
    os.makedirs(LABDIR, exist_ok=True)
    os.makedirs(os.path.join(wb.BUILD, "lab"), exist_ok=True)
    cc = wb.hipcc()
    others = [os.path.join(wb.BUILD, s.replace(".hip", ".o")) for s in wb.SOURCES if s != "gemm.hip"]
    only = os.environ.get("VARIANTS")  # VARIANTS=a,b: build / run only these
    for name, flags in VARIANTS.items():
        if only and name not in only.split(","):
            continue
        obj = os.path.join(wb.BUILD, "lab", f"{name}.o")
        subprocess.run([cc] + wb.COMMON + ["-DWF_GEMM_TIMING"] + flags + ["-c", os.path.join(wb.CSRC, "gemm.hip"), "-o", obj], check=True)
        lib = os.path.join(LABDIR, f"libwf_hip_{name}.so")
        subprocess.run([cc, "-shared", "-fPIC", f"--offload-arch={wb.ARCH}", "-o", lib, obj] + others + ["-L/opt/rocm/lib", "-lamdhip64"], check=True)
        print(lib)


def child():
    import torch
    from worldforge_amd import _ffi, dit
    lib = _ffi.lib()
    buf = (ctypes.c_ulonglong * 16)()
    print("| shape M x N x K | epilogue | ms | TFLOP/s | tiles | prologue | K loop (group A / B) | epilogue | epilogue share | MFMA pipe cycles of the K loop | K-loop efficiency | clock GHz (busy cycles per CU / wall) |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    short = len(sys.argv) > 2 and sys.argv[2] == "short"
    p1 = len(sys.argv) > 2 and sys.argv[2] == "p1"
    for P in ((1,) if short or p1 else (1, 8)):
        M = (32760 + P - 1) // P
        for (N, K, epi, name) in ((15360, 5120, 0, "bf16"),) if short else ((15360, 5120, 0, "bf16"), (5120, 5120, 3, "x += gate * y"), (14080, 5120, 1, "bf16 + GELU"), (5120, 13824, 3, "x += gate * y")):
            x = torch.randn(M, K, device="cuda").bfloat16()
            w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).bfloat16()
            b, g = torch.randn(N, device="cuda"), torch.randn(N, device="cuda")
            out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16 if epi < 2 else torch.float32)
            fn = lambda: dit.gemm(x, w, b, out, epi, gate=g if epi == 3 else None)  # noqa: E731
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            lib.wf_debug_gemm_cycles(buf, 1)
            e0.record()
            for _ in range(10):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            lib.wf_debug_gemm_cycles(buf, 1)
            n = max(buf[11], 1)
            pro, kl, ep, klb = buf[8] / n, buf[9] / n, buf[10] / n, buf[12] / n
            tiles = n // 10
            # MFMA pipe cycles one SIMD spends on a tile's K loop: two waves per SIMD, each (wave tile MFMAs) x 32 cycles
            wide = N % 320 == 0
            mfma_per_wave = (5 * 2 if wide else 2 * 4) * 4 * (K // 64)
            pipe = 2 * mfma_per_wave * 32
            rounds = -(-tiles // 256)
            busy = (pro + kl + ep) * rounds   # cycles a CU is busy per launch if every round costs the same
            print(f"| {M} x {N} x {K} | {name} | {ms:.3f} | {2.0 * M * N * K / ms / 1e9:.0f} | {tiles} ({tiles / 256:.2f} rounds) | {pro:.0f} | {kl:.0f} / {klb:.0f} | {ep:.0f} | "
                  f"{100 * ep / (pro + kl + ep):.1f} % (+ prologue {100 * pro / (pro + kl + ep):.1f} %) | {pipe} | {100 * pipe / kl:.1f} % | {busy / (ms * 1e-3) / 1e9:.2f} |", flush=True)


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else "run"
    if mode == "build":
        build()
    elif mode == "child":
        child()
    elif mode == "pick":     # the full shape table for the variants named in VARIANTS=a,b (e.g. the touch-prefetch experiment of the RMW epilogue)
        for name in os.environ["VARIANTS"].split(","):
            print(f"\n== {name} {VARIANTS[name]}", flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), "child", "p1"], env=dict(os.environ, WF_LIB=os.path.join(LABDIR, f"libwf_hip_{name}.so")), timeout=900)
    elif mode == "pick":     # the full shape table (1 GPU's token count) for the variants named in VARIANTS=a,b
        for name in os.environ["VARIANTS"].split(","):
            print(f"\n== {name} {VARIANTS[name]}", flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), "child", "p1"], env=dict(os.environ, WF_LIB=os.path.join(LABDIR, f"libwf_hip_{name}.so")), timeout=900)
    elif mode == "ablate":   # K-loop cycles of every ablation variant on the QKV shape only
        for name in VARIANTS:
            print(f"\n== {name} {VARIANTS[name]}", flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), "child", "short"], env=dict(os.environ, WF_LIB=os.path.join(LABDIR, f"libwf_hip_{name}.so")), timeout=600)
    else:
        sys.exit(subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, WF_LIB=LIB)).returncode)
