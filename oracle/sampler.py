"""Oracle: the IRR / FLF / DSG guided sampling loop (torch CPU).  TEST INFRASTRUCTURE ONLY.

Restates WanImageToVideoPipeline.prepare_latents (PIPE:301-362) and the loop of __call__ (PIPE:562-744) of
/root/reference/wan_for_worldforge/utils/pipeline_wan_i2v_clean.py, with the scheduler of oracle/sched.py.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Optional

import torch

from . import inject, sched


@dataclass
class SamplerConfig:
    num_inference_steps: int = 50
    guidance_scale: float = 5.0
    flow_shift: float = 3.0
    guided: bool = False
    resample_steps: int = 1
    guide_steps: int = 20
    omega: float = 1.8
    omega_resample: float = 1.0
    resample_round: int = 20
    use_pca_channel_selection: bool = False
    flow_backend: str = "tdiff"
    transformer_dtype: torch.dtype = torch.bfloat16


def prepare_condition(image: torch.Tensor, num_frames: int, encode_mode: Callable, mean, std, t_scale: int = 4):
    """PIPE:327-362: first-frame conditioning latents + 4-channel temporal mask -> [B,20,T,h,w] fp32.
    image: [B,3,H,W] in [-1,1]."""
    B, _, H, W = image.shape
    vid = torch.cat([image.unsqueeze(2), image.new_zeros(B, 3, num_frames - 1, H, W)], dim=2).to(torch.float32)
    lat = inject.latent_norm(encode_mode(vid), mean, std, torch.float32)
    h, w = lat.shape[-2:]
    m = torch.ones(B, 1, num_frames, h, w)
    m[:, :, 1:] = 0
    first = torch.repeat_interleave(m[:, :, 0:1], dim=2, repeats=t_scale)
    m = torch.concat([first, m[:, :, 1:]], dim=2)
    m = m.view(B, -1, t_scale, h, w).transpose(1, 2)
    return torch.concat([m, lat], dim=1)


def run(cfg: SamplerConfig, *, latents: torch.Tensor, condition: torch.Tensor, transformer: Callable,
        prompt_embeds, negative_prompt_embeds, image_embeds, video_ref: Optional[torch.Tensor],
        mask: Optional[torch.Tensor], decode: Callable, encode_mode: Callable, mean, std,
        generator: Optional[torch.Generator] = None, trace: Optional[list] = None) -> torch.Tensor:
    """PIPE:515-728.  Returns the final latents (normalised space).  `trace` collects per-step records."""
    st = sched.make_state(cfg.num_inference_steps, cfg.flow_shift)
    do_cfg = cfg.guidance_scale > 1
    omega = cfg.omega
    for i, t in enumerate(st.timesteps):
        st.derivative_history = []
        pred_x0 = None
        out_prev = None
        for r in range(cfg.resample_steps):
            if r > 0:
                sched.set_resample_mode(st, True)
                t_model = sched.get_resample_timestep(st, i).expand(latents.shape[0])
                st.step_index -= 1
                if st.lower_order_nums > 0 and st.last_lower_order_nums < st.solver_order:
                    st.lower_order_nums -= 1
                st.this_order = st.last_this_order
            else:
                sched.set_resample_mode(st, False)
                t_model = t.expand(latents.shape[0])
            x_in = torch.cat([latents, condition], dim=1).to(cfg.transformer_dtype)
            v = transformer(x_in, t_model, prompt_embeds, image_embeds)
            if do_cfg:
                v_un = transformer(x_in, t_model, negative_prompt_embeds, image_embeds)
                v = inject.cfg_combine(v, v_un, cfg.guidance_scale)
                if r < 1:
                    st.derivative_history.append(v)
            guided_now = cfg.guided and i < cfg.guide_steps and video_ref is not None
            fuse = None
            if guided_now and mask is not None:
                def fuse(x0, _r=r):
                    return inject.fuse_latents(x0, video_ref, mask, decode=decode, encode_mode=encode_mode, mean=mean,
                                               std=std, use_flf=cfg.use_pca_channel_selection, resampling=_r > 0,
                                               current_step=i, flow_backend=cfg.flow_backend)
            out_prev, pred_x0 = sched.step(st, v, t, latents, fuse=fuse, resampling=r > 0,
                                           is_resample_round=i < cfg.resample_round, resample_count=cfg.resample_steps)
            if trace is not None:
                trace.append(("step", i, r, out_prev.clone(), pred_x0.clone()))
            if i >= cfg.resample_round:
                break
            if r < cfg.resample_steps - 1:
                if generator is not None:
                    noise = torch.randn(pred_x0.shape, generator=generator)
                else:
                    noise = torch.randn(pred_x0.shape)
                t_noise = sched.get_resample_timestep(st, i)
                latents = sched.add_noise_resample(st, pred_x0, noise, t_noise)
        if len(st.derivative_history) > 1:
            good, worse = st.derivative_history[-1], st.derivative_history[0]
            if i >= cfg.guide_steps:
                omega = cfg.omega_resample
            better = inject.dsg(good, worse, omega)
            st.step_index -= 1
            if st.lower_order_nums > 0 and st.last_lower_order_nums < st.solver_order:
                st.lower_order_nums -= 1
            x0b = sched.convert_model_output(st, better, latents)
            st.last_sample = latents
            st.model_outputs[-1] = x0b
            latents = sched.unip_update(st, latents, st.this_order)
            st.step_index += 1
            if 0 <= st.lower_order_nums < st.solver_order:
                st.lower_order_nums += 1
            latents = latents.to(cfg.transformer_dtype)
        else:
            latents = out_prev
        sched.set_resample_mode(st, False)
        if trace is not None:
            trace.append(("latents", i, latents.clone()))
    return latents


def decode_final(latents: torch.Tensor, decode: Callable, mean, std):
    """PIPE:732-744: de-normalise (fp32), decode, (x/2+0.5).clamp(0,1) -> [B,F,H,W,C]."""
    z = inject.latent_denorm(latents.to(torch.float32), mean, std)
    video = decode(z)
    return (video / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 4, 1)
