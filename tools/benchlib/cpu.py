"""bench.py's `cpu_baseline` leg: the ONLY place of the bench that imports oracle/ (as the thing timed beside the GPU number, never as the product)."""
from __future__ import annotations

import os
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _median3(fn):
    """median of 3 runs after 1 warm-up."""
    fn()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[1]


def cpu_baseline(L, frames, height, width):
    """BASELINE.md section 3 on a bounded sample (~15-25 s on the GPU box's host): the oracle (CPU port of the reference arithmetic,
    fp32, all host threads), median of 3 after a warm-up of
      (i)   one full-width DiT block (d = 5120, 40 heads, FFN 13824, text+image cross-attention) at the C1 token count L1 = 4524, and its
            self-attention core alone at L1 -- everything in a block except that core is linear in L;
      (ii)  the self-attention core with the config's TRUE key length: L1 query rows x L keys on 4 of the 40 heads (the core is linear
            in query rows and in heads, so this prices the L^2 term at the real L without the 40 x L x L score tensor);
      (iii) VAE encode + decode of a 5 x 96 x 96 clip (linear in pixel-frames);
    returns per-unit CPU seconds at the config's true sizes: one DiT forward (40 blocks) and one VAE decode + encode."""
    from oracle import dit as odit
    from oracle import vae as ovae

    torch.manual_seed(0)
    cores = torch.get_num_threads()
    cfg = odit.DiTConfig(num_layers=1)
    L1 = 4524  # BASELINE config 1: 9 frames of 464 x 832
    W = odit.random_weights(cfg, seed=1)
    f, h, w = 3, 29, 52
    tok = torch.randn(L1, cfg.dim)
    e0 = torch.randn(6, cfg.dim) * 0.1
    ctx = torch.randn(769, cfg.dim)
    ang = odit.rope_tables(128, f, h, w)
    nh, hd = cfg.num_heads, cfg.dim // cfg.num_heads
    q1, k1, v1 = (torch.randn(L1, nh, hd) for _ in range(3))
    hs = 4
    qL, kL, vL = torch.randn(L1, hs, hd), torch.randn(L, hs, hd), torch.randn(L, hs, hd)
    with torch.no_grad():
        t_blk = _median3(lambda: odit.block(tok, e0, ctx, W, 0, cfg, ang))
        t_core1 = _median3(lambda: odit.attention(q1, k1, v1))
        t_coreL = _median3(lambda: odit.attention(qL, kL, vL))
    del W
    t_block_true = (t_blk - t_core1) * (L / L1) + t_coreL * (nh / hs) * (L / L1)
    Wv = ovae.random_weights(seed=2)
    Fs, Hs, Ws = 5, 96, 96
    xs = torch.rand(1, 3, Fs, Hs, Ws) * 2 - 1
    with torch.no_grad():
        t_vae = _median3(lambda: ovae.decode(Wv, ovae.encode_mode(Wv, xs)))
    t_vae_true = t_vae * (frames * height * width) / (Fs * Hs * Ws)
    return dict(cores=cores, t_dit_forward_s=40 * t_block_true, t_vae_roundtrip_s=t_vae_true,
                sample=f"oracle fp32, {cores} threads, median of 3 after warm-up: DiT block (d=5120, 40 heads, FFN 13824) at L1={L1}: {t_blk:.2f}s "
                       f"(its self-attention core {t_core1:.2f}s); core with the true key length {L1} q x {L} k on {hs}/40 heads: {t_coreL:.2f}s; "
                       f"VAE encode+decode {Fs}x{Hs}x{Ws}: {t_vae:.2f}s; block@L = (block - core)*L/L1 + core_L*(40/{hs})*L/L1 = "
                       f"{t_block_true:.1f}s, x40 blocks per forward; VAE scaled by pixel-frames to {frames}x{height}x{width}: {t_vae_true:.0f}s; "
                       "steps/s = steps / sum(count x unit time) over the timed step mix (extrapolation)")


def cpu_baseline_longcat():
    """Oracle (CPU port, fp32) timed on this host: one LongCat block at the released width on a bounded token sample + the VAE sample
    of cpu_baseline(); returns flop rates."""
    from oracle import longcat_dit as olc
    from oracle import vae as ovae

    torch.manual_seed(0)
    cfg = olc.LongCatConfig(depth=1)
    W = olc.random_weights(cfg, seed=1)
    T, h, w = 2, 32, 32  # 512 tokens
    Ls = T * (h // 2) * (w // 2)
    x, cap = torch.randn(16, T, h, w), torch.randn(64, cfg.caption_channels)
    with torch.no_grad():
        t0 = time.time()
        olc.forward(W, cfg, x, torch.tensor([0.0, 500.0]), cap, None, num_cond_latents=1)
        t_blk = time.time() - t0
    C, Hd = cfg.hidden_size, cfg.ffn_hidden
    flop_blk = 2.0 * Ls * C * (6 * C + 3 * Hd) + 4.0 * Ls * Ls * C
    del W
    Wv = ovae.random_weights(seed=2)
    Fs, Hs, Ws = 5, 64, 64
    with torch.no_grad():
        t0 = time.time()
        ovae.decode(Wv, ovae.encode_mode(Wv, torch.rand(1, 3, Fs, Hs, Ws) * 2 - 1))
        t_vae = time.time() - t0
    return dict(cores=torch.get_num_threads(), dit_flops_per_s=flop_blk / t_blk, vae_flops_per_s=(5.19e6 + 8.70e6) * Fs * Hs * Ws / t_vae,
                sample=f"oracle fp32: 1 LongCat block (d=4096, 32 heads, SwiGLU 11008) + embeddings at L={Ls} tokens in {t_blk:.2f}s + VAE "
                       f"encode+decode of {Fs}x{Hs}x{Ws} in {t_vae:.2f}s; extrapolated by algorithmic FLOPs to the timed step mix")
