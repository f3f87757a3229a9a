"""LongCat-Video DiT forward at the 480p / 93-frame configuration (latent 24 x 60 x 104 -> 37 440 tokens, 48 blocks, 13.6 B
parameters, random weights): ms per forward and the self-attention rate.  python tools/longcat_bench.py [--depth N] [--iters K]"""
import argparse
import sys
import time

import torch

sys.path.insert(0, ".")
from worldforge_amd import dit  # noqa: E402
from worldforge_amd.longcat_dit import LongCatConfig, LongCatVideoTransformer3DModel  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--depth", type=int, default=48)
    ap.add_argument("--iters", type=int, default=2)
    ap.add_argument("--frames", type=int, default=24)
    ap.add_argument("--h", type=int, default=60)
    ap.add_argument("--w", type=int, default=104)
    ap.add_argument("--bsa", action="store_true", help="refine-pass configuration: block-sparse self-attention, 4 condition latents, no CFG")
    ap.add_argument("--refine", action="store_true", help="time the 720p refine pass (704 x 1280, 93 stage-1 frames): prepare, 2 steps")
    ap.add_argument("--job", action="store_true", help="time guided / plain sampler steps of the whole i2v job instead of one forward")
    a = ap.parse_args()
    if a.refine:
        from worldforge_amd.longcat_pipeline import LongCatVideoPipeline
        from worldforge_amd.longcat_scheduler import FlowMatchEulerDiscreteScheduler
        from worldforge_amd.vae import AutoencoderKLWan

        dev = torch.device("cuda:0")
        m = LongCatVideoTransformer3DModel(LongCatConfig(depth=a.depth), dev, enable_bsa=True).init_random(1)
        vae = AutoencoderKLWan(dev).init_random(seed=1)
        pipe = LongCatVideoPipeline(vae, FlowMatchEulerDiscreteScheduler(shift=12.0), m, device=dev)
        g = torch.Generator().manual_seed(3)
        frames = (torch.rand(93, 480, 832, 3, generator=g) * 255).to(torch.uint8)
        image = torch.rand(3, 704, 1280, generator=g)
        pe = (torch.randn(1, 1, 512, 4096, generator=g) * 0.5).bfloat16()
        pm = torch.zeros(1, 512, dtype=torch.int64)
        pm[:, :180] = 1
        marks = [("t0", "", time.time())]

        class _Stop(Exception):
            pass

        def hook(i, what):
            torch.cuda.synchronize()
            marks.append((i, what, time.time()))
            if what == "end" and i == 1:
                raise _Stop

        try:
            pipe.generate_refine(stage1_video=frames, height=704, width=1280, prompt_embeds=pe, prompt_attention_mask=pm, image=image,
                                 num_cond_frames=1, num_inference_steps=50, generator=torch.manual_seed(1), t_thresh=0.5,
                                 spatial_refine_only=True, step_hook=hook)
        except _Stop:
            pass
        t = {(i, w): ts for i, w, ts in marks}
        print(f"refine 704x1280, 93 frames (109 after padding = 28 latent frames = 98 560 tokens), {len(pipe.scheduler.timesteps)} steps "
              f"at t_thresh 0.5: prepare (up-sample + 2 VAE encodes + noise mix) {t[(0, 'start')] - t[('t0', '')]:.2f} s; "
              f"step 0 {t[(0, 'end')] - t[(0, 'start')]:.2f} s, step 1 {t[(1, 'end')] - t[(1, 'start')]:.2f} s", flush=True)
        sys.exit(0)
    if a.job:
        from worldforge_amd.longcat_pipeline import LongCatVideoPipeline
        from worldforge_amd.longcat_scheduler import FlowMatchEulerDiscreteScheduler
        from worldforge_amd.vae import AutoencoderKLWan

        dev = torch.device("cuda:0")
        m = LongCatVideoTransformer3DModel(LongCatConfig(depth=a.depth), dev).init_random(1)
        vae = AutoencoderKLWan(dev).init_random(seed=1)
        pipe = LongCatVideoPipeline(vae, FlowMatchEulerDiscreteScheduler(shift=12.0), m, device=dev)
        Fr, H, Wd = 4 * (a.frames - 1) + 1, a.h * 8, a.w * 8
        g = torch.Generator().manual_seed(3)
        image = torch.rand(3, H, Wd, generator=g)
        ref = torch.rand(1, 3, Fr, H, Wd, generator=g)
        mask = (torch.rand(1, 1, Fr, H // 8, Wd // 8, generator=g) > 0.4).float().repeat_interleave(8, 3).repeat_interleave(8, 4)
        pe, ne = (torch.randn(2, 1, 1, 512, 4096, generator=g) * 0.5).bfloat16()
        pm, nm = torch.zeros(1, 512, dtype=torch.int64), torch.zeros(1, 512, dtype=torch.int64)
        pm[:, :180] = 1
        nm[:, :120] = 1
        marks = []

        class _Stop(Exception):
            pass

        def hook(i, what):
            torch.cuda.synchronize()
            marks.append((i, what, time.time()))
            if what == "end" and i == 3:
                raise _Stop

        try:
            pipe.generate_i2v(image=image, height=H, width=Wd, prompt_embeds=pe, prompt_attention_mask=pm, negative_prompt_embeds=ne,
                              negative_prompt_attention_mask=nm, num_frames=Fr, num_inference_steps=50, guidance_scale=4.0,
                              generator=torch.manual_seed(1), video_ref=ref, mask=mask, guided=True, resample_steps=3, guide_steps=3,
                              resample_round=3, use_pca_channel_selection=True, static=True, step_hook=hook)
        except _Stop:
            pass
        t = {(i, w): ts for i, w, ts in marks}
        for i in range(4):
            print(f"step {i} ({'guided: 3 CFG evaluations + injection + DSG' if i < 3 else 'plain: 1 CFG evaluation'}): "
                  f"{t[(i, 'end')] - t[(i, 'start')]:.2f} s", flush=True)
        sys.exit(0)
    dev = torch.device("cuda:0")
    cfg = LongCatConfig(depth=a.depth)
    t0 = time.time()
    m = LongCatVideoTransformer3DModel(cfg, dev, enable_bsa=a.bsa).init_random(1)
    torch.cuda.synchronize()
    print(f"init {time.time() - t0:.1f}s, {m.param_bytes() / 1e9:.1f} GB of weights", flush=True)
    g = torch.Generator(device=dev).manual_seed(2)
    x = torch.randn((16, a.frames, a.h, a.w), generator=g, device=dev).bfloat16()
    cap = torch.randn((512, cfg.caption_channels), generator=g, device=dev).bfloat16()
    mask = torch.zeros(512, dtype=torch.int64)
    mask[:180] = 1
    ncl = 4 if a.bsa else 1
    ts = [0.0] * ncl + [700.0] * (a.frames - ncl)
    if a.bsa:
        import worldforge_amd.bsa as wbsa
        ev = []
        orig = wbsa.sparse_attention

        def timed(*aa, **kk):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = orig(*aa, **kk)
            e1.record()
            ev.append((e0, e1, aa[0].shape[1], kk.get("n_k_blocks", aa[6] if len(aa) > 6 else 0)))
            return r

        wbsa.sparse_attention = timed
        orig_t = wbsa.sparse_attention_topk

        def timed_topk(*aa, **kk):  # (fused selection + list building + the sparse kernel: both launches inside the events)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = orig_t(*aa, **kk)
            e1.record()
            ev.append((e0, e1, aa[0].shape[1], aa[4].shape[2]))
            return r

        wbsa.sparse_attention_topk = timed_topk
    m.forward_tokens(x, ts, cap, mask, ncl)
    torch.cuda.synchronize()
    dit.PROFILE_ATTN = []
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    if a.bsa:
        ev.clear()
    for _ in range(a.iters):
        out = m.forward_tokens(x, ts, cap, mask, ncl)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    L = a.frames * (a.h // 2) * (a.w // 2)
    tpf = (a.h // 2) * (a.w // 2)
    C, Hd = cfg.hidden_size, cfg.ffn_hidden
    gemm_flop = 2.0 * L * C * (3 * C + C + C + C + 3 * Hd) * a.depth
    attn_flop = 4.0 * (L - tpf) * L * C * a.depth
    if a.bsa:
        big = [(s.elapsed_time(e), lq, nk) for s, e, lq, nk in ev if lq > 4 * tpf]
        ms_b = sum(b[0] for b in big) / len(big)
        lq, nk = big[0][1], big[0][2]
        nsel = int(0.125 * nk)
        print(f"tokens {L}, forward {ms:.1f} ms; noise-token block-sparse attention {ms_b:.2f} ms per launch: {lq // 128} query blocks x {nsel} of {nk} "
              f"key blocks = {4.0 * lq * nsel * 128 * C / ms_b / 1e9:.0f} TFLOP/s of selected work (dense would be {4.0 * lq * nk * 128 * C / 1e12:.0f} TFLOP); "
              f"finite={bool(torch.isfinite(out).all())}")
        sys.exit(0)
    att = [s.elapsed_time(e) for s, e in dit.PROFILE_ATTN]
    att_ms = sum(att) / len(att)
    print(f"tokens {L}, forward {ms:.1f} ms ({(gemm_flop + attn_flop) / ms / 1e9:.0f} TFLOP/s end to end); noise-token self-attention "
          f"{att_ms:.2f} ms = {4.0 * (L - tpf) * L * C / att_ms / 1e9:.0f} TFLOP/s; finite={bool(torch.isfinite(out).all())}")
