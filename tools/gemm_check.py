"""A-B check of wf_gemm_bf16 on the shapes the VAE really calls it with: harvest (M, N, K, ldx, ldw, ldo, epilogue) from one decode + encode at the
480p latent grid (unsharded and as rank 3 of 8), then run every distinct shape on seeded operands `reps` times and print one SHA-1 per run.
Run under two libraries (WF_LIB=...) and diff the lines."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from worldforge_amd import vae as wvae, dit

DEV = "cuda:0"


def harvest():
    seen = []
    real = dit.gemm

    def spy(x, w, bias, out, epi, gate=None):
        key = (x.shape[0], w.shape[0], x.shape[1], x.stride(0), w.stride(0), out.stride(0), epi, bias is not None)
        if key not in seen:
            seen.append(key)
        return real(x, w, bias, out, epi, gate)
    wvae.gemm = spy
    m = wvae.AutoencoderKLWan(DEV, precision="bf16").init_random(seed=1)
    g = torch.Generator(device=DEV).manual_seed(0)
    z = torch.randn((1, 16, 21, 60, 104), generator=g, device=DEV)
    video = torch.rand((1, 3, 81, 480, 832), generator=g, device=DEV) * 2 - 1
    m.decode(z, return_dict=False)
    m.encode(video)
    wvae.gemm = real
    return seen


def run(key, reps=4):
    M, N, K, ldx, ldw, ldo, epi, hb = key
    g = torch.Generator(device=DEV).manual_seed(M * 7 + N * 3 + K)
    x = torch.randn((M, ldx), generator=g, device=DEV).bfloat16()[:, :K]
    w = (torch.randn((N, ldw), generator=g, device=DEV) / K ** 0.5).bfloat16()[:, :K]
    b = torch.randn((N,), generator=g, device=DEV) if hb else None
    odt = torch.bfloat16 if epi in (0, 1) else torch.float32
    hs = []
    for _ in range(reps):
        out = torch.zeros((M, ldo), dtype=odt, device=DEV)[:, :N]
        dit.gemm(x, w, b, out, epi)
        torch.cuda.synchronize()
        hs.append(hashlib.sha1(out.contiguous().view(torch.uint8).cpu().numpy().tobytes()).hexdigest()[:12])
    print(f"M={M} N={N} K={K} ldx={ldx} ldw={ldw} ldo={ldo} epi={epi} bias={int(hb)}: {' '.join(hs)}{'' if len(set(hs)) == 1 else '  NON-DETERMINISTIC'}", flush=True)


if __name__ == "__main__":
    for key in harvest():
        run(key)
