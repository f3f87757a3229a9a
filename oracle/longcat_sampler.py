"""Oracle: the LongCat-Video guided image-to-video sampling loop (torch CPU).  TEST INFRASTRUCTURE ONLY.

Restates, from /root/reference/longcat_for_worldforge/longcat_video:
  pipeline_longcat_video.py (PIPE)  : get_timesteps_sigmas :317-331, prepare_latents :214-286, optimized_scale :374-383,
                                      the loop of generate_i2v :823-994, the decode tail :998-1004
  modules/scheduling_flow_match_euler_discrete.py (SCHED): set_timesteps :610-716, step :740-912, add_noise :1041-1070,
                                      fuse_latents :1072-1233, the channel selector :165-170, 172-243, 245-381: its temporal-difference
                                      branch (executed when `import cv2` fails, SCHED:45-51, 307-309; the golden fixtures were
                                      recorded through it) and its Farneback branch (:105-163, 290-300; cv2 restated by
                                      oracle/farneback.py: PARITY UNPINNED for that branch)
Pinned against trajectories recorded from the imported, unmodified reference pipeline + scheduler with deterministic stand-ins for
the DiT / VAE / text encoder (tests/golden/g12_longcat_pipe_*.npz, tools/make_goldens.py longcat_pipe).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence

import numpy as np
import torch

from . import inject


@dataclass
class LongCatSamplerConfig:
    num_inference_steps: int = 50
    guidance_scale: float = 4.0
    shift: float = 1.0               # scheduler config (SCHED:447)
    use_distill: bool = False
    guided: bool = False
    resample_steps: int = 3
    guide_steps: int = 20
    resample_round: int = 20
    omega: float = 1.8
    omega_resample: float = 1.0
    use_pca_channel_selection: bool = False
    max_replace_threshold: Optional[int] = None
    dit_dtype: torch.dtype = torch.bfloat16
    vae_dtype: torch.dtype = torch.float32   # `vae.dtype`; the LongCat entry loads the VAE in bf16 (run_longcat_worldforge_single.py:205)
    flow_backend: str = "tdiff"      # "farneback" = the branch taken where cv2 imports (SCHED:45-51)


# ---- schedule ---------------------------------------------------------------------------------------------------------------
def timesteps_sigmas(sampling_steps: int, use_distill: bool = False, num_timesteps: int = 1000, num_distill: int = 50) -> torch.Tensor:
    """PIPE:317-331."""
    if use_distill:
        idx = torch.arange(1, num_distill + 1, dtype=torch.float32)
        idx = (idx * (num_timesteps // num_distill)).round().long()
        inf = np.floor(np.linspace(0, num_distill, num=sampling_steps, endpoint=False)).astype(np.int64)
        sigmas = torch.flip(idx, [0])[inf].float() / num_timesteps
        sigmas = sigmas - sigmas[-1]
    else:
        sigmas = torch.linspace(0.999, 0.000, sampling_steps)
    return sigmas.to(torch.float32)


def make_schedule(sigmas_in: torch.Tensor, shift: float, num_train_timesteps: int = 1000):
    """SCHED:664-709 for the static-shift configuration: sigmas [n+1] (terminal 0 appended), timesteps [n], fp32."""
    s = (sigmas_in.numpy() if isinstance(sigmas_in, torch.Tensor) else np.array(sigmas_in)).astype(np.float32)
    s = shift * s / (1 + (shift - 1) * s)
    sig = torch.from_numpy(s).to(torch.float32)
    return torch.cat([sig, torch.zeros(1)]), sig * num_train_timesteps


# ---- latents ----------------------------------------------------------------------------------------------------------------
def normalize(z, mean, std):
    """PIPE:385-394: (z - mean) * (1 / std) with the constants in z's dtype."""
    m = torch.tensor(mean).view(1, -1, 1, 1, 1).to(z.device, z.dtype)
    s = 1.0 / torch.tensor(std).view(1, -1, 1, 1, 1).to(z.device, z.dtype)
    return (z - m) * s


def denormalize(z, mean, std):
    """PIPE:396-405."""
    m = torch.tensor(mean).view(1, -1, 1, 1, 1).to(z.device, z.dtype)
    s = 1.0 / torch.tensor(std).view(1, -1, 1, 1, 1).to(z.device, z.dtype)
    return z / s + m


def prepare_latents(image: torch.Tensor, num_frames: int, encode_sample: Callable, mean, std, generator, dit_dtype=torch.bfloat16):
    """PIPE:774-792 + 236-286 with num_cond_frames = 1: noise latents (fp32, drawn first), first latent frame <- normalised posterior
    SAMPLE of the conditioning image (cast to the DiT dtype first, PIPE:772).  image [B,3,H,W] in [-1,1]."""
    B, _, H, W = image.shape
    image = image.to(dit_dtype)
    T = (num_frames - 1) // 4 + 1
    latents = torch.randn((B, 16, T, H // 8, W // 8), generator=generator).to(torch.float32)
    cond = torch.cat([encode_sample(image[i].unsqueeze(0).unsqueeze(2), generator) for i in range(B)], dim=0).to(torch.float32)
    latents[:, :, :1] = normalize(cond, mean, std)
    return latents


# ---- FLF, LongCat variant ---------------------------------------------------------------------------------------------------
def flow_similarity(ref_motion: torch.Tensor, cand_motion: torch.Tensor) -> float:
    """SCHED:172-243: like the Wan metric but outliers are epe > 3 OR epe > 5 % of |ref|, and the weights are 0.4 / 0.4 / 0.2."""
    n = min(ref_motion.shape[1], cand_motion.shape[1])
    if n <= 0:
        return 0.0
    r, c = ref_motion.float()[:, :n], cand_motion.float()[:, :n]
    r = r[:, :, :2] if r.shape[2] >= 2 else r.repeat(1, 1, 2, 1, 1)[:, :, :2]
    c = c[:, :, :2] if c.shape[2] >= 2 else c.repeat(1, 1, 2, 1, 1)[:, :, :2]
    d = r - c
    epe = torch.sqrt((d ** 2).sum(dim=2) + 1e-8)
    dot = (r * c).sum(dim=2)
    rn = torch.sqrt((r ** 2).sum(dim=2) + 1e-8)
    cn = torch.sqrt((c ** 2).sum(dim=2) + 1e-8)
    ae = torch.acos(torch.clamp(dot / (rn * cn + 1e-8), -1.0, 1.0)) * 180.0 / torch.pi
    outl = (epe > 3.0) | (epe > rn * 0.05)
    m_epe, m_ae, fl = epe.mean(), ae.mean(), outl.float().mean()
    werr = 0.4 * torch.clamp(m_epe / 10.0, 0.0, 1.0) + 0.4 * torch.clamp(fl / 0.5, 0.0, 1.0) + 0.2 * torch.clamp(m_ae / 30.0, 0.0, 1.0)
    return torch.clamp(1.0 - werr, 0.0, 1.0).item()


def farneback_motion(channel: torch.Tensor) -> torch.Tensor:
    """SCHED:290-292 + 105-158 for one channel [1, 1, T, h, w] -> [1, T-1, 2, h, w] fp32.  The three RGB planes are equal, so one is
    carried; normalisation runs in the tensor's dtype (bf16 ops round after every step), the uint8 step in numpy float32."""
    from . import farneback

    n = (channel - channel.min()) / (channel.max() - channel.min() + 1e-8)
    v = n.to(torch.float32)[0, 0].numpy()
    if v.min() >= -1.1 and v.max() <= 1.1:  # always: n lies in [0, 1]
        q = ((v + 1.0) * 127.5).clip(0, 255).astype(np.uint8)
    else:
        q = (v * 255).clip(0, 255).astype(np.uint8)
    fl = [farneback.calc_optical_flow_farneback(q[t], q[t + 1]) for t in range(q.shape[0] - 1)]
    return torch.from_numpy(np.stack(fl, axis=0).transpose(0, 3, 1, 2).astype(np.float32)).unsqueeze(0)


def channel_similarities(pred: torch.Tensor, enc: torch.Tensor, flow_backend: str = "tdiff") -> List[float]:
    """SCHED:281-316: temporal-difference branch (SCHED:165-170) or the Farneback branch (SCHED:286-300)."""
    enc = enc.to(pred.device, dtype=pred.dtype)
    sims = []
    if flow_backend == "farneback":
        for c in range(pred.shape[1]):
            sims.append(flow_similarity(farneback_motion(enc[:, c:c + 1]), farneback_motion(pred[:, c:c + 1])))
        return sims
    assert flow_backend == "tdiff", flow_backend
    for c in range(pred.shape[1]):
        pm = inject.temporal_diff_motion(pred[:, c:c + 1].to(torch.float32))
        rm = inject.temporal_diff_motion(enc[:, c:c + 1].to(torch.float32))
        sims.append(flow_similarity(rm, pm))
    return sims


def select_from_similarities(sims: Sequence[float], current_step: int, use_distill: bool = False,
                             max_replace_threshold: Optional[int] = None) -> List[int]:
    """SCHED:269-270, 330-381."""
    if current_step < 2:
        return []
    corr = np.array(sims)
    mean, std = np.mean(corr), np.std(corr)
    early = 3 if use_distill else 5
    if current_step <= early:
        chans = np.argsort(corr)[:1].tolist()
    else:
        max_replace = max_replace_threshold if max_replace_threshold is not None else (3 if use_distill else 1)
        thr = mean - 0.625 * std
        below = [i for i, s in enumerate(corr) if s < thr]
        if len(below) < 1:
            chans = np.argsort(corr)[:1].tolist()
        elif len(below) > max_replace:
            sc = sorted([(i, corr[i]) for i in below], key=lambda x: x[1])
            chans = [i for i, _ in sc[:max_replace]]
        else:
            chans = below
    return sorted(chans)


def fuse_latents(x0_full: torch.Tensor, ref: torch.Tensor, mask: torch.Tensor, *, decode: Callable, encode_mode: Callable, mean, std,
                 use_flf: bool, current_step: int, use_distill: bool = False, max_replace_threshold: Optional[int] = None,
                 flow_backend: str = "tdiff", vae_dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """SCHED:1072-1233: de-normalise -> decode -> ref*mask + dec*(1-mask) at the exact decoded size (no alignment: a shape mismatch
    raises inside the reference's try block and returns the prediction unchanged) -> encode (mode) -> normalise -> FLF.
    vae_dtype: the two `.to(dtype=vae.dtype)` hand-offs (SCHED:1124, 1166); the blend runs in the decoded video's dtype (:1152-1164) and
    the normalisation promotes the encoded latents back to the prediction's dtype (:1183, constants in that dtype :1113-1120)."""
    dec = decode(denormalize(x0_full, mean, std).to(vae_dtype))
    if ref.shape != dec.shape or mask.shape[1] != 1 or tuple(mask.shape[2:]) != tuple(dec.shape[2:]):
        return x0_full
    r = 2.0 * ref.to(dec.dtype) - 1.0
    m = mask.to(dec.dtype)
    if m.shape[1] != dec.shape[1]:
        m = m.repeat(1, dec.shape[1], 1, 1, 1)
    fused = (r * m + dec * (1 - m)).to(vae_dtype)
    enc = encode_mode(fused)
    if enc.shape != x0_full.shape:
        return x0_full
    m_ = torch.tensor(mean).view(1, -1, 1, 1, 1).to(x0_full.device, x0_full.dtype)
    s_ = 1.0 / torch.tensor(std).view(1, -1, 1, 1, 1).to(x0_full.device, x0_full.dtype)
    enc = (enc - m_) * s_   # = normalize(enc) for an fp32 VAE; a bf16 `enc` is promoted by the fp32 constants
    if use_flf:
        for c in select_from_similarities(channel_similarities(x0_full, enc, flow_backend), current_step, use_distill, max_replace_threshold):
            if 0 <= c < enc.shape[1]:
                enc[:, c] = x0_full[:, c]
    return enc.to(x0_full.dtype)


# ---- CFG-zero ---------------------------------------------------------------------------------------------------------------
def cfg_zero(cond: torch.Tensor, uncond: torch.Tensor, g: float) -> torch.Tensor:
    """PIPE:374-383 + 875-885: st* = <cond, uncond> / (|uncond|^2 + 1e-8) per sample; uncond*st* + g*(cond - uncond*st*)."""
    B = cond.shape[0]
    pos, neg = cond.reshape(B, -1), uncond.reshape(B, -1)
    st = (torch.sum(pos * neg, dim=1, keepdim=True) / (torch.sum(neg ** 2, dim=1, keepdim=True) + 1e-8)).view(B, 1, 1, 1, 1)
    return uncond * st + g * (cond - uncond * st)


# ---- the loop ---------------------------------------------------------------------------------------------------------------
def run(cfg: LongCatSamplerConfig, *, latents: torch.Tensor, dit: Callable, prompt_embeds: torch.Tensor, prompt_mask: torch.Tensor,
        video_ref: Optional[torch.Tensor], mask: Optional[torch.Tensor], decode: Callable, encode_mode: Callable, mean, std,
        generator: Optional[torch.Generator] = None, trace: Optional[list] = None) -> torch.Tensor:
    """PIPE:823-994.  latents [B,16,T,h,w] fp32 with the clean condition frame at index 0 (edited in place, as the reference);
    prompt_embeds [2B,1,N,C] = (negative, positive) when guidance_scale > 1, else [B,1,N,C]; dit(hidden_states, timestep[B,T],
    encoder_hidden_states, encoder_attention_mask, num_cond_latents) -> fp32."""
    sigmas, timesteps = make_schedule(timesteps_sigmas(cfg.num_inference_steps, cfg.use_distill), cfg.shift)
    do_cfg = cfg.guidance_scale > 1.0
    for i, t in enumerate(timesteps):
        history = []
        sigma, dt = sigmas[i], sigmas[i + 1] - sigmas[i]
        pred_x0 = None
        prev = None

        def euler(model_output, sample, fuse):
            x0 = sample - sigma * model_output  # SCHED:836
            if fuse is not None:
                x0_full = latents - sigma * torch.cat([torch.zeros_like(model_output[:, :, 0:1]), model_output], dim=2)
                x0 = fuse(x0_full)[:, :, 1:]
            history.append(model_output)
            return (sample + dt * model_output).to(model_output.dtype), x0  # SCHED:894, 901

        rounds = cfg.resample_steps if (cfg.guided and i < cfg.resample_round) else 1
        for r in range(rounds):
            ts = t.expand(latents.shape[0]).to(cfg.dit_dtype)
            x_in = (torch.cat([latents] * 2) if do_cfg else latents).to(cfg.dit_dtype)
            if do_cfg:
                ts = torch.cat([ts] * 2)
            ts = ts.unsqueeze(-1).repeat(1, x_in.shape[2])
            ts[:, :1] = 0
            v = dit(hidden_states=x_in, timestep=ts, encoder_hidden_states=prompt_embeds, encoder_attention_mask=prompt_mask,
                    num_cond_latents=1)
            if do_cfg:
                v_un, v_c = v.chunk(2)
                v = cfg_zero(v_c, v_un, cfg.guidance_scale)
            v = -v
            fuse = None
            if cfg.guided and i < cfg.guide_steps and video_ref is not None and r == 0:
                def fuse(x0_full):
                    return fuse_latents(x0_full, video_ref, mask, decode=decode, encode_mode=encode_mode, mean=mean, std=std,
                                        use_flf=cfg.use_pca_channel_selection, current_step=i, use_distill=cfg.use_distill,
                                        max_replace_threshold=cfg.max_replace_threshold, flow_backend=cfg.flow_backend,
                                        vae_dtype=cfg.vae_dtype)
            prev, pred_x0 = euler(v[:, :, 1:], latents[:, :, 1:].to(torch.float32), fuse)
            if trace is not None:
                trace.append(("step", i, r, prev.clone(), pred_x0.clone()))
            if i >= cfg.resample_round:
                break
            if r < cfg.resample_steps - 1 and pred_x0 is not None:
                noise = torch.randn(pred_x0.shape, generator=generator).to(dtype=pred_x0.dtype)
                latents[:, :, 1:] = (1.0 - sigma) * pred_x0 + sigma * noise  # SCHED:1041-1070 at the current timestep
        if i < cfg.resample_round and len(history) > 1 and cfg.guided:
            omega = cfg.omega_resample if i >= cfg.guide_steps else cfg.omega
            better = inject.dsg(history[-1], history[0], omega)
            latents[:, :, 1:] = euler(better, latents[:, :, 1:].to(torch.float32), None)[0]
        else:
            latents[:, :, 1:] = prev
        if trace is not None:
            trace.append(("latents", i, latents.clone()))
    return latents


def decode_final(latents: torch.Tensor, decode: Callable, mean, std, vae_dtype: torch.dtype = torch.float32):
    """PIPE:998-1004: latents.to(vae.dtype), de-normalise IN THAT DTYPE, decode, (x/2+0.5).clamp(0,1) in the video's dtype
    (diffusers VideoProcessor.denormalize) -> float -> [B,F,H,W,C]."""
    video = decode(denormalize(latents.to(vae_dtype), mean, std))
    return (video / 2 + 0.5).clamp(0, 1).float().permute(0, 2, 3, 4, 1)


# ---- 720p refine pass (PIPE:1271-1511) --------------------------------------------------------------------------------------
def refine_upsample(stage1_frames: torch.Tensor, height: int, width: int, new_frame_size: int, dtype=torch.bfloat16,
                    gpu_semantics: bool = False) -> torch.Tensor:
    """PIPE:1404-1413: uint8 frames [F, H0, W0, 3] -> [1, 3, new_frame_size, height, width] in [-1, 1], every step in `dtype`.
    gpu_semantics=False runs torch's CPU bf16 interpolation kernels (what the golden recorded here contains: they round intermediate
    passes to bf16, up to 1.5 grey levels off); gpu_semantics=True restates what the reference executes on a GPU, where torch's
    upsample kernels interpolate in fp32 (accscalar_t) and round once per op -- the semantics the HIP kernel implements."""
    import torch.nn.functional as F
    if gpu_semantics:
        rb = lambda x: x.to(dtype).float()  # noqa: E731
        v = stage1_frames.permute(0, 3, 1, 2).float()
        down = rb(F.interpolate(v, size=(height, width), mode="bilinear", align_corners=True))
        down = rb(down.permute(1, 0, 2, 3).unsqueeze(0) / 255.0)
        up = rb(F.interpolate(down, size=(new_frame_size, height, width), mode="trilinear", align_corners=True))
        return rb(rb(up * 2) - 1).to(dtype)
    v = stage1_frames.permute(0, 3, 1, 2).to(dtype)
    down = F.interpolate(v, size=(height, width), mode="bilinear", align_corners=True)
    down = down.permute(1, 0, 2, 3).unsqueeze(0) / 255.0
    up = F.interpolate(down, size=(new_frame_size, height, width), mode="trilinear", align_corners=True)
    return up * 2 - 1


def refine_plan(n_frames_up: int, num_cond_frames: int, t_scale: int = 4, granularity: int = 4):
    """PIPE:1415-1431: latent-frame counts padded to the block-sparse granularity -> (num_cond_latents, num_cond_frames_added,
    num_noise_frames_added, num_cond_frames after padding)."""
    import math
    num_noise_frames = n_frames_up - num_cond_frames
    ncl = added_c = 0
    if num_cond_frames > 0:
        ncl = 1 + math.ceil((num_cond_frames - 1) / t_scale)
        ncl = math.ceil(ncl / granularity) * granularity
        added_c = 1 + (ncl - 1) * t_scale - num_cond_frames
        num_cond_frames = num_cond_frames + added_c
    nnl = math.ceil(math.ceil(num_noise_frames / t_scale) / granularity) * granularity
    return ncl, added_c, nnl * t_scale - num_noise_frames, num_cond_frames


def refine_schedule(num_inference_steps: int, shift: float, t_thresh: float):
    """PIPE:1394-1402: the usual schedule, cut to [t_thresh * 1000] + the timesteps below it; sigmas = timesteps / 1000 + [0]."""
    _, ts = make_schedule(timesteps_sigmas(num_inference_steps), shift)
    if t_thresh:
        tt = torch.tensor(t_thresh * 1000, dtype=ts.dtype)
        ts = torch.cat([tt.unsqueeze(0), ts[ts < tt]])
    return torch.cat([ts / 1000, torch.zeros(1)]), ts


def run_refine(*, stage1_frames: torch.Tensor, image: torch.Tensor, height: int, width: int, dit: Callable, prompt_embeds, prompt_mask,
               encode_sample: Callable, decode: Callable, mean, std, generator, num_inference_steps: int = 50, shift: float = 1.0,
               t_thresh: float = 0.5, spatial_refine_only: bool = False, num_cond_frames: int = 1, dit_dtype=torch.bfloat16,
               trace: Optional[list] = None, gpu_upsample: bool = False, vae_dtype: torch.dtype = torch.float32):
    """PIPE:1394-1503 for one sample with an image condition.  image [1,3,H,W] in [-1,1] or None.  Returns frames [1,F,H,W,3]."""
    sigmas, timesteps = refine_schedule(num_inference_steps, shift, t_thresh)
    nf = stage1_frames.shape[0]
    new_frame_size = nf if spatial_refine_only else 2 * nf
    up = refine_upsample(stage1_frames, height, width, new_frame_size, dit_dtype, gpu_semantics=gpu_upsample)
    ncl, added_c, added_n, ncf = refine_plan(up.shape[2], num_cond_frames if image is not None else 0)
    up = torch.cat([up[:, :, 0:1].repeat(1, 1, added_c, 1, 1), up, up[:, :, -1:].repeat(1, 1, added_n, 1, 1)], dim=2)
    lat = normalize(encode_sample(up, generator), mean, std)
    noise = torch.randn(lat.shape, generator=generator, dtype=lat.dtype)
    latents = ((1 - t_thresh) * lat + t_thresh * noise).to(torch.float32)
    if image is not None:  # PIPE:262-284
        enc_in = image.to(dit_dtype)[0].unsqueeze(0).unsqueeze(2)
        enc_in = torch.cat([enc_in[:, :, 0:1].repeat(1, 1, added_c, 1, 1), enc_in], dim=2)
        assert enc_in.shape[2] == ncf
        latents[:, :, :ncl] = normalize(encode_sample(enc_in, generator).to(torch.float32), mean, std)
    for i, t in enumerate(timesteps):
        ts = t.expand(latents.shape[0]).to(dit_dtype).unsqueeze(-1).repeat(1, latents.shape[2])
        ts[:, :ncl] = 0
        v = -dit(hidden_states=latents.to(dit_dtype), timestep=ts, encoder_hidden_states=prompt_embeds, encoder_attention_mask=prompt_mask,
                 num_cond_latents=ncl)
        latents[:, :, ncl:] = (latents[:, :, ncl:].to(torch.float32) + (sigmas[i + 1] - sigmas[i]) * v[:, :, ncl:]).to(v.dtype)
        if trace is not None:
            trace.append(latents.clone())
    frames = decode_final(latents, decode, mean, std, vae_dtype)
    return frames[:, added_c: new_frame_size + added_c]
