"""Short driver-timed windows of BASELINE configs 3 and 4 (and of config 2 with the TF32-class VAE) after the headline window."""
from __future__ import annotations

import os
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from .launcher import progress
from .measure import _attn_frac, synthetic_inputs


LONGCAT_VAE_NOTE = ("bf16 module, bf16 matrix operands, fp32 accumulate and an f32 stream inside: the dtype the reference entry loads its VAE in "
                    "(longcat_for_worldforge/run_longcat_worldforge_single.py:205, TF32 allowed :144-146), with the `.to(vae.dtype)` hand-offs of "
                    "fuse_latents (scheduling_flow_match_euler_discrete.py:1124, 1166) pinned on reference trajectories (tests/golden/g12b, g15b); "
                    "0.8e-2 / 1.1e-2 rel. L2 from the fp32 network where the reference's eager bf16 module is at 1.5e-2 / 2.1e-2 (g8c)")


def also_wan_tf32_vae(pipe, device, guided_ms_default, plain_ms_default, frames=81, H=480, W=832):
    """The headline workload (C2) with the TF32-CLASS VAE (precision "fp16": one fp16 term per operand, what an fp32 cuDNN convolution
    keeps of its multiplicands under PyTorch's default torch.backends.cudnn.allow_tf32 = True) instead of the headline's fp32-class
    three-term operands: steps 13..16 of the 50-step schedule = one guided warm-up step, then 1 guided + 2 plain.  NOT the headline
    configuration (BASELINE.md states an fp32 VAE); a labelled extra line.  The PSNR figures are __graft_entry__.smoke()'s job (HIP
    sampler vs the fp32 CPU oracle, the checker) run with either VAE precision."""
    from worldforge_amd.vae import AutoencoderKLWan
    guide = 15
    image, ref, mask, text, neg, img_emb = synthetic_inputs(frames, H, W, device)
    vae0 = pipe.vae
    pipe.vae = AutoencoderKLWan(device, precision="fp16").init_random(seed=1)
    marks = {}

    def hook(i, phase):
        torch.cuda.synchronize()
        marks[(phase[0], i)] = time.perf_counter()
        if phase == "begin":
            progress(f"also: C2 with the TF32-class VAE, step {i}")

    try:
        pipe(image=image, height=H, width=W, num_frames=frames, num_inference_steps=50, guidance_scale=4.0, generator=torch.manual_seed(42),
             prompt_embeds=text, negative_prompt_embeds=neg, image_embeds=img_emb, output_type="latent", video_ref=ref, mask=mask, guided=True,
             resample_steps=2, guide_steps=guide, omega=4.0, omega_resample=4.0, resample_round=guide, use_pca_channel_selection=True,
             static=True, start_step=guide - 2, max_steps=4, step_hook=hook)
    finally:
        pipe.vae = vae0
    g = 1e3 * (marks[("e", guide - 1)] - marks[("b", guide - 1)])
    pl = [1e3 * (marks[("e", i)] - marks[("b", i)]) for i in (guide, guide + 1)]
    p = sum(pl) / len(pl)
    out = {"workload": f"Wan2.1-I2V-14B-480P, {frames}f {H}x{W}, 50-step schedule, full IRR+FLF+DSG, CFG 4 -- the headline workload with the "
                       "TF32-class VAE; timed steps 14..16 = 1 guided + 2 plain after one guided warm-up step",
           "vae_precision": "fp16 (ONE fp16 term per operand: 10 explicit mantissa bits = TF32's, fp32 accumulate, f32 stream; 1.0e-3 / 1.4e-3 rel. L2 "
                            "of mu / decode from the fp32 goldens; the headline line above runs fp16x3)",
           "steps_per_s": 50.0 / ((15 * g + 35 * p) / 1e3), "steps_per_s_basis": "the 50-step job's 15 guided : 35 plain mix of the timed step times",
           "guided_step_ms": g, "plain_step_ms": p,
           "headline_same_run": {"guided_step_ms": guided_ms_default, "plain_step_ms": plain_ms_default,
                                 "steps_per_s": (50.0 / ((15 * guided_ms_default + 35 * plain_ms_default) / 1e3)
                                                 if guided_ms_default and plain_ms_default else None)}}
    import __graft_entry__ as ge   # the smoke job: product on this GPU against the fp32 CPU oracle (checker use only)
    (p3, _), (p1, _) = ge.parity_run(dim=256, ffn_dim=512, heads=2, layers=2, Fr=9, H=32, Wd=32, steps=3, guide=2, vae_precision=("fp16x3", "fp16"))
    out["psnr_db_vs_fp32_oracle_smoke_job"] = {"fp16x3 (headline)": p3, "fp16 (this line)": p1, "bar": 40.0}
    return out


def also_wan_720p(pipe, model, cfg, device, frames=81):
    """BASELINE config 3 (Wan2.1-I2V-14B-720P, 81 frames, full IRR + FLF + DSG, CFG 4) on the resident 14B model: steps 14, 15, 16 of the
    50-step schedule = 1 guided + 2 plain, after ONE guided warm-up step (step 13: the 720p buffers' first touch, round 5 timed it)."""
    from worldforge_amd import dit as wdit
    H, W, guide = 720, 1280, 15
    image, ref, mask, text, neg, img_emb = synthetic_inputs(frames, H, W, device)
    marks = {}

    def hook(i, phase):
        torch.cuda.synchronize()
        marks[(phase[0], i)] = time.perf_counter()
        if phase == "begin" and i == guide - 1:
            wdit.PROFILE_ATTN = []
        if phase == "begin":
            progress(f"also: 720p step {i}")

    pipe(image=image, height=H, width=W, num_frames=frames, num_inference_steps=50, guidance_scale=4.0, generator=torch.manual_seed(42),
         prompt_embeds=text, negative_prompt_embeds=neg, image_embeds=img_emb, output_type="latent", video_ref=ref, mask=mask, guided=True,
         resample_steps=2, guide_steps=guide, omega=4.0, omega_resample=4.0, resample_round=guide, use_pca_channel_selection=True,
         static=True, start_step=guide - 2, max_steps=4, step_hook=hook)
    L = ((frames - 1) // 4 + 1) * (H // 16) * (W // 16)
    frac, avg = _attn_frac(wdit, 4.0 * L * L * 128 * cfg.num_heads)
    g = 1e3 * (marks[("e", guide - 1)] - marks[("b", guide - 1)])
    pl = [1e3 * (marks[("e", i)] - marks[("b", i)]) for i in (guide, guide + 1)]
    p = sum(pl) / len(pl)
    return {"workload": f"Wan2.1-I2V-14B-720P, {frames}f {H}x{W}, 50-step schedule, full IRR+FLF+DSG, CFG 4; timed steps 14..16 = 1 guided + 2 plain after one guided warm-up step",
            "tokens": L, "steps_per_s": 50.0 / ((15 * g + 35 * p) / 1e3), "steps_per_s_basis": "the 50-step job's 15 guided : 35 plain mix of the timed step times",
            "guided_step_ms": g, "plain_step_ms": p, "attn_frac": frac, "attn_avg_launch_ms": avg}


def also_longcat(device, height=480, width=832, frames=93):
    """BASELINE config 4 (LongCat-Video distilled 480p, 16 steps + the 720p refine pass) on a random-init 13.6 B model: steps 1..3 of the
    distilled 16-step schedule (1 guided step = 3 IRR rounds + FLF + DSG, 2 plain; no CFG) after one guided warm-up step, then steps 0 and 1
    of the 704 x 1280 refine pass (block-sparse self-attention at 98 560 tokens; step 1 reported)."""
    from worldforge_amd import dit as wdit
    from worldforge_amd.longcat_dit import LongCatConfig, LongCatVideoTransformer3DModel
    from worldforge_amd.longcat_pipeline import LongCatVideoPipeline
    from worldforge_amd.longcat_scheduler import FlowMatchEulerDiscreteScheduler
    from worldforge_amd.vae import AutoencoderKLWan

    cfg = LongCatConfig()
    model = LongCatVideoTransformer3DModel(cfg, device).init_random(seed=0)
    # the VAE as the LongCat entry loads it: AutoencoderKLWan.from_pretrained(..., torch_dtype=torch.bfloat16) (run_longcat_worldforge_single.py:205)
    vae = AutoencoderKLWan(device, precision="bf16", dtype=torch.bfloat16).init_random(seed=1)
    pipe = LongCatVideoPipeline(vae, FlowMatchEulerDiscreteScheduler(shift=12.0), model, device=device)
    g = torch.Generator().manual_seed(42)
    image = torch.rand(3, height, width, generator=g)
    ref = torch.rand(1, 3, frames, height, width, generator=g)
    mask = (torch.rand(1, 1, frames, height // 8, width // 8, generator=g) > 0.4).float().repeat_interleave(8, 3).repeat_interleave(8, 4)
    pe, ne = (torch.randn(2, 1, 1, 512, cfg.caption_channels, generator=g) * 0.5).bfloat16()
    pm, nm = torch.zeros(1, 512, dtype=torch.int64), torch.zeros(1, 512, dtype=torch.int64)
    pm[:, :180] = 1
    nm[:, :120] = 1
    marks = {}

    class _Stop(Exception):
        pass

    def hook(i, phase):
        torch.cuda.synchronize()
        marks[(phase[0], i)] = time.perf_counter()
        if phase == "start" and i == 1:
            wdit.PROFILE_ATTN = []
        if phase == "end" and i == 3:
            raise _Stop

    try:   # steps 0 and 1 guided, 2 and 3 plain; step 0 is the warm-up (a cold pipeline's first guided step measured 1-1.5 s long)
        pipe.generate_i2v(image=image, height=height, width=width, prompt_embeds=pe, prompt_attention_mask=pm, negative_prompt_embeds=ne,
                          negative_prompt_attention_mask=nm, num_frames=frames, num_inference_steps=16, use_distill=True, guidance_scale=1.0,
                          generator=torch.manual_seed(42), output_type="latent", video_ref=ref, mask=mask, guided=True, resample_steps=3,
                          guide_steps=2, resample_round=2, omega=1.8, omega_resample=1.0, use_pca_channel_selection=True, static=True,
                          step_hook=hook)
    except _Stop:
        pass
    T = (frames - 1) // 4 + 1
    tpf = (height // 16) * (width // 16)
    L = T * tpf
    frac, avg = _attn_frac(wdit, 4.0 * (L - tpf) * L * 128 * cfg.num_heads)
    gms = 1e3 * (marks[("e", 1)] - marks[("s", 1)])
    pms = sum(1e3 * (marks[("e", i)] - marks[("s", i)]) for i in (2, 3)) / 2
    out = {"workload": f"LongCat-Video 13.6B distilled i2v, {frames}f {height}x{width}, 16-step schedule, IRR x3 + FLF + DSG, no CFG; timed steps 1..3 = "
                       "1 guided + 2 plain after one guided warm-up step",
           "tokens": L, "vae_precision": LONGCAT_VAE_NOTE,
           "steps_per_s": 16.0 / ((6 * gms + 10 * pms) / 1e3), "steps_per_s_basis": "the 16-step job's 6 guided : 10 plain mix of the timed step times",
           "guided_step_ms": gms, "plain_step_ms": pms, "attn_frac": frac, "attn_avg_launch_ms": avg}
    # ---- the 720p refine pass (pipeline_longcat_video.py:1271-1511) on the same weights with block-sparse self-attention
    model._ws.clear()
    torch.cuda.empty_cache()
    model.enable_bsa()
    stage1 = (torch.rand(frames, height, width, 3, generator=g) * 255).to(torch.uint8)
    image2 = torch.rand(3, 704, 1280, generator=g)
    rm = {"t0": time.perf_counter()}

    def rhook(i, what):
        torch.cuda.synchronize()
        rm[(what[0], i)] = time.perf_counter()
        if what == "end" and i == 1:
            raise _Stop

    try:
        pipe.generate_refine(stage1_video=stage1, height=704, width=1280, prompt_embeds=pe, prompt_attention_mask=pm, image=image2,
                             num_cond_frames=1, num_inference_steps=50, generator=torch.manual_seed(1), t_thresh=0.5,
                             spatial_refine_only=True, step_hook=rhook)
    except _Stop:
        pass
    out["refine_720p"] = {"workload": "generate_refine 704x1280, 93 stage-1 frames -> 28 latent frames = 98 560 tokens, block-sparse self-attention "
                                      "(sparsity 0.875), no CFG, t_thresh 0.5; steps 0 and 1 timed, step 1 reported",
                          "prepare_s": rm[("s", 0)] - rm["t0"], "step_ms": 1e3 * (rm[("e", 1)] - rm[("s", 1)]),
                          "first_step_ms": 1e3 * (rm[("e", 0)] - rm[("s", 0)])}
    return out
