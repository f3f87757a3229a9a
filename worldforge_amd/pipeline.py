"""The IRR / FLF / DSG guided sampler, HIP-backed.

Host-side mirror of `WanImageToVideoPipeline` (PIPE = /root/reference/wan_for_worldforge/utils/pipeline_wan_i2v_clean.py):
same call signature for the sampling knobs (PIPE:390-424), same control flow (PIPE:562-728), same RNG draw order (CPU
generator, PIPE:323 and :644), same dtype hand-offs (latents become bf16 after a DSG step, PIPE:708).  The transformer,
the VAE and the scheduler are passed in as objects speaking the reference's (diffusers') call protocol, so the
MI355X-native modules of this package (dit.WanTransformer3DModel, vae.AutoencoderKLWan, scheduler.UniPCMultistepScheduler)
and test doubles are interchangeable.  Text / image encoders run once per video outside the loop and are out of scope
(SURVEY 2 #6): their outputs are taken as tensors (`prompt_embeds`, `negative_prompt_embeds`, `image_embeds`).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Callable, Dict, List, Optional, Union

import numpy as np
import torch

from . import ops, trace


@dataclass
class WanPipelineOutput:
    frames: Any


class WanImageToVideoPipeline:
    def __init__(self, transformer, vae, scheduler, device: Union[str, torch.device] = "cuda:0"):
        self.transformer = transformer
        self.vae = vae
        self.scheduler = scheduler
        self.device = torch.device(device)
        tds = getattr(vae, "temperal_downsample", [False, True, True])
        self.vae_scale_factor_temporal = 2 ** sum(tds)  # PIPE:162
        self.vae_scale_factor_spatial = 2 ** len(tds)   # PIPE:163
        self._guidance_scale = 1.0
        self.tracer = trace.Tracer.from_env()  # WF_TRACE=<file.json>: roctx ranges + per-phase GPU time log (trace.py)
        self.timing: Dict[str, Dict[str, float]] = {}  # {phase: {count, total_ms, mean_ms}} of the last call when tracing is on

    @property
    def guidance_scale(self):
        return self._guidance_scale

    @property
    def do_classifier_free_guidance(self):
        return self._guidance_scale > 1

    # ---- PIPE:262-299 ---------------------------------------------------------------------------------------
    def check_inputs(self, image, height, width, prompt_embeds, negative_prompt_embeds, image_embeds):
        if image is None:
            raise ValueError("Must provide `image` (the first frame) -- it conditions the latents (PIPE:327-351)")
        if not isinstance(image, torch.Tensor) and not hasattr(image, "resize"):
            raise ValueError(f"`image` must be torch.Tensor or PIL.Image, got {type(image)}")
        if height % 16 != 0 or width % 16 != 0:
            raise ValueError(f"`height` and `width` must be divisible by 16, got {height} and {width}")
        if prompt_embeds is None:
            raise ValueError("Must provide `prompt_embeds` (text encoder output [B,512,4096]); the encoder is out of scope")
        if image_embeds is None:
            raise ValueError("Must provide `image_embeds` (CLIP penultimate hidden states [B,257,1280])")

    def _preprocess_image(self, image, height, width) -> torch.Tensor:
        """diffusers VideoProcessor.preprocess: -> [1,3,H,W] fp32 in [-1,1] on the device."""
        if isinstance(image, torch.Tensor):
            t = image if image.dim() == 4 else image.unsqueeze(0)
            t = t.to(torch.float32)
        else:
            if image.size != (width, height):
                image = image.resize((width, height))
            t = torch.from_numpy(np.array(image).astype(np.float32) / 255.0).permute(2, 0, 1).unsqueeze(0)
        if t.shape[-2:] != (height, width):
            raise ValueError(f"image tensor is {tuple(t.shape[-2:])}, expected {(height, width)}")
        return (2.0 * t - 1.0).to(self.device)

    # ---- PIPE:301-362 ---------------------------------------------------------------------------------------
    def prepare_latents(self, image: torch.Tensor, batch_size: int, num_channels_latents: int, height: int, width: int,
                        num_frames: int, generator=None, latents: Optional[torch.Tensor] = None):
        ts, ss = self.vae_scale_factor_temporal, self.vae_scale_factor_spatial
        T = (num_frames - 1) // ts + 1
        h, w = height // ss, width // ss
        shape = (batch_size, num_channels_latents, T, h, w)
        if latents is None:
            # diffusers randn_tensor: CPU generator -> draw on CPU, then move (RNG parity with the reference)
            if generator is not None and generator.device.type == "cpu":
                latents = torch.randn(shape, generator=generator, dtype=torch.float32).to(self.device)
            else:
                latents = torch.randn(shape, generator=generator, dtype=torch.float32, device=self.device)
        else:
            latents = latents.to(device=self.device, dtype=torch.float32)
        video_condition = torch.zeros((image.shape[0], image.shape[1], num_frames, height, width), dtype=torch.float32,
                                      device=self.device)
        video_condition[:, :, 0] = image
        mu = self.vae.encode(video_condition).latent_dist.mode()
        latent_condition = ops.latent_norm(mu, self.vae.config.latents_mean, self.vae.config.latents_std)
        if batch_size != latent_condition.shape[0]:
            latent_condition = latent_condition.repeat(batch_size, 1, 1, 1, 1)
        # PIPE:353-360: first latent frame is "given" in all 4 temporal sub-slots
        mask_lat = torch.zeros((batch_size, ts, T, h, w), dtype=torch.float32, device=self.device)
        mask_lat[:, :, 0] = 1.0
        return latents, torch.cat([mask_lat, latent_condition], dim=1)

    # ---- PIPE:388-753 ---------------------------------------------------------------------------------------
    @torch.no_grad()
    def __call__(self, image, prompt=None, negative_prompt=None, height: int = 480, width: int = 832, num_frames: int = 81,
                 num_inference_steps: int = 50, guidance_scale: float = 5.0, num_videos_per_prompt: int = 1,
                 generator: Optional[torch.Generator] = None, latents: Optional[torch.Tensor] = None,
                 prompt_embeds: Optional[torch.Tensor] = None, negative_prompt_embeds: Optional[torch.Tensor] = None,
                 image_embeds: Optional[torch.Tensor] = None, output_type: str = "np", return_dict: bool = True,
                 attention_kwargs=None, callback_on_step_end: Optional[Callable] = None,
                 callback_on_step_end_tensor_inputs: List[str] = ["latents"], max_sequence_length: int = 512,
                 video_ref: Optional[torch.Tensor] = None, mask=None, guided: bool = False, resample_steps: int = 1,
                 guide_steps: int = 20, omega: float = 1.8, omega_resample: float = 1.0, resample_round: int = 20,
                 use_pca_channel_selection: bool = False, static: bool = False, start_step: int = 0,
                 max_steps: Optional[int] = None, step_hook: Optional[Callable] = None):
        """Extra (non-reference) arguments: `start_step` / `max_steps` run a window of the schedule (used by bench.py);
        `step_hook(i, phase)` is called at outer-step boundaries for timing."""
        if prompt is not None:
            raise ValueError("text encoding is out of scope: pass `prompt_embeds` / `negative_prompt_embeds`")
        self.check_inputs(image, height, width, prompt_embeds, negative_prompt_embeds, image_embeds)
        ts = self.vae_scale_factor_temporal
        if num_frames % ts != 1:
            num_frames = num_frames // ts * ts + 1  # PIPE:475-477
        num_frames = max(num_frames, 1)
        self._guidance_scale = guidance_scale
        device = self.device
        sch = self.scheduler
        batch_size = prompt_embeds.shape[0]
        transformer_dtype = self.transformer.dtype
        prompt_embeds = prompt_embeds.to(device=device, dtype=transformer_dtype)
        if self.do_classifier_free_guidance:
            if negative_prompt_embeds is None:
                raise ValueError("classifier-free guidance needs `negative_prompt_embeds`")
            negative_prompt_embeds = negative_prompt_embeds.to(device=device, dtype=transformer_dtype)
        image_embeds = image_embeds.to(device=device, dtype=transformer_dtype)
        if image_embeds.shape[0] != batch_size:
            image_embeds = image_embeds.repeat(batch_size, 1, 1)

        sch.set_timesteps(num_inference_steps, device=device)
        timesteps = sch.timesteps
        img = self._preprocess_image(image, height, width)
        latents, condition = self.prepare_latents(img, batch_size * num_videos_per_prompt, self.vae.config.z_dim, height,
                                                  width, num_frames, generator, latents)
        if video_ref is not None and guided:
            if not isinstance(video_ref, torch.Tensor):
                video_ref = torch.tensor(video_ref)
            video_ref = video_ref.to(dtype=torch.float32).to(device=device)
        if mask is not None and guided:
            if isinstance(mask, np.ndarray):
                mask = torch.from_numpy(mask)
            if mask.dim() == 3:
                mask = mask.unsqueeze(0).unsqueeze(1)
            elif mask.dim() == 4 and mask.shape[1] != 1:
                mask = mask[:, 0:1, :, :].unsqueeze(0)
            elif mask.dim() == 4 and mask.shape[0] == 1:
                mask = mask.unsqueeze(1)
            elif mask.dim() == 5 and mask.shape[1] != 1:
                mask = mask[:, 0:1, :, :, :]
            mask = mask.to(device=device, dtype=torch.float32).contiguous()  # fp64 -> fp32 is what SCHED:1374 does per call
        if not hasattr(sch, "derivative_history"):
            sch.derivative_history = []

        tr = self.tracer
        tr.reset()  # `self.timing` and the WF_TRACE log describe THIS call only
        sch.tracer = tr
        if start_step:
            # bench window: enter the schedule at `start_step` (UniPC restarts at order 1 there, like step 0)
            sch._step_index = start_step
        n_done = 0
        for i, t in enumerate(timesteps):
            if i < start_step:
                continue
            if max_steps is not None and n_done >= max_steps:
                break
            if step_hook is not None:
                step_hook(i, "begin")
            sch.derivative_history = []
            pred_original_sample = None
            scheduler_output = None
            for r in range(resample_steps):
                if r > 0:
                    sch.set_resample_mode(True)
                    timestep_for_transformer = sch.get_resample_timestep(i).expand(latents.shape[0])
                else:
                    sch.set_resample_mode(False)
                    timestep_for_transformer = t.expand(latents.shape[0])
                if r > 0:
                    sch._step_index -= 1
                    if sch.lower_order_nums > 0 and sch.last_lower_order_nums < sch.config.solver_order:
                        sch.lower_order_nums -= 1
                    sch.this_order = sch.last_this_order
                latent_model_input = self._model_input(latents, condition, transformer_dtype)
                with tr.range("dit_cfg", step=i, round=r):
                    pair = getattr(self.transformer, "forward_cfg_pair", None) if self.do_classifier_free_guidance else None
                    split = getattr(self, "cfg_split", None) if self.do_classifier_free_guidance else None
                    if split is not None:
                        # CFG groups x sequence shards (SURVEY 8e "P = 8 = 2 x 4"; parallel.Comm.split): the ranks of group 0 run the
                        # positive-prompt forward, those of group 1 the negative one -- ONE forward per rank and evaluation on a token
                        # shard twice as long, half the K / V^T exchange -- and every rank then takes both velocities from one small
                        # all-gather over the whole job (a rank's slot holds its group's full tensor: [P, ...] -> slots 0 and P / 2)
                        world_comm, branch = split
                        own = self.transformer(hidden_states=latent_model_input, timestep=timestep_for_transformer,
                                               encoder_hidden_states=prompt_embeds if branch == 0 else negative_prompt_embeds,
                                               encoder_hidden_states_image=image_embeds, attention_kwargs=attention_kwargs,
                                               return_dict=False)[0].contiguous()
                        both = torch.empty((world_comm.world,) + tuple(own.shape), dtype=own.dtype, device=own.device)
                        world_comm.all_gather(both, own)
                        noise_pred, noise_uncond = both[0], both[world_comm.world // 2]
                    elif pair is not None:
                        # same two calls as below, advanced in lock-step so that under sequence parallelism each branch's K / V
                        # exchange overlaps the other branch's compute (dit.forward_tokens_pair); identical results
                        noise_pred, noise_uncond = pair(hidden_states=latent_model_input, timestep=timestep_for_transformer,
                                                        encoder_hidden_states=prompt_embeds,
                                                        negative_encoder_hidden_states=negative_prompt_embeds,
                                                        encoder_hidden_states_image=image_embeds)
                    else:
                        noise_pred = self.transformer(hidden_states=latent_model_input, timestep=timestep_for_transformer,
                                                      encoder_hidden_states=prompt_embeds,
                                                      encoder_hidden_states_image=image_embeds, attention_kwargs=attention_kwargs,
                                                      return_dict=False)[0]
                        if self.do_classifier_free_guidance:
                            noise_uncond = self.transformer(hidden_states=latent_model_input, timestep=timestep_for_transformer,
                                                            encoder_hidden_states=negative_prompt_embeds,
                                                            encoder_hidden_states_image=image_embeds,
                                                            attention_kwargs=attention_kwargs, return_dict=False)[0]
                    if self.do_classifier_free_guidance:
                        noise_pred = ops.cfg_combine(noise_pred, noise_uncond, guidance_scale)
                        if r < 1:
                            sch.derivative_history.append(noise_pred)
                with tr.range("scheduler_step", step=i, round=r):
                    scheduler_output = sch.step(noise_pred, t, latents, mask=mask,
                                                guided=guided and i < guide_steps and r < resample_steps,
                                                video_latents=video_ref, vae=self.vae, resampling=r > 0, return_dict=True,
                                                current_step=i, resample_count=resample_steps,
                                                is_resample_round=i < resample_round,
                                                use_pca_channel_selection=use_pca_channel_selection, static=static)
                pred_original_sample = scheduler_output.pred_x0
                if i >= resample_round:
                    break
                if r < resample_steps - 1 and pred_original_sample is not None:
                    if generator is not None:
                        noise = torch.randn(pred_original_sample.shape, generator=generator).to(device=device)
                    else:
                        noise = torch.randn(pred_original_sample.shape, device=device)
                    with tr.range("renoise", step=i, round=r):
                        latents = sch.add_noise(pred_original_sample, noise, sch.get_resample_timestep(i), r,
                                                use_resample_sigma=True)
            if len(sch.derivative_history) > 1:
                noise_pred_good = sch.derivative_history[-1]
                noise_pred_worse = sch.derivative_history[0]
                if i >= guide_steps:
                    omega = omega_resample
                with tr.range("dsg", step=i):
                    noise_pred_better = ops.dsg(noise_pred_good, noise_pred_worse, omega)
                sch._step_index -= 1
                if sch.lower_order_nums > 0 and sch.last_lower_order_nums < sch.config.solver_order:
                    sch.lower_order_nums -= 1
                noise_pred_convert = sch.convert_model_output(noise_pred_better, sample=latents)
                sch.last_sample = latents
                sch.model_outputs[-1] = noise_pred_convert
                latents = sch.multistep_uni_p_bh_update(model_output=noise_pred_better, sample=latents, order=sch.this_order)
                sch._step_index += 1
                if 0 <= sch.lower_order_nums < sch.config.solver_order:
                    sch.lower_order_nums += 1
                latents = ops.cast(latents, transformer_dtype)
            else:
                latents = scheduler_output.prev_sample
            sch.set_resample_mode(False)
            if callback_on_step_end is not None:
                cb_out = callback_on_step_end(self, i, t, {"latents": latents}) or {}
                latents = cb_out.pop("latents", latents)
            n_done += 1
            if step_hook is not None:
                step_hook(i, "end")

        if output_type == "latent":
            video = latents
        else:
            z = ops.latent_denorm(ops.cast(latents, torch.float32), self.vae.config.latents_mean,
                                  self.vae.config.latents_std)
            with tr.range("final_decode"):
                video = self.vae.decode(z, return_dict=False)[0]
                video = torch.stack([ops.postprocess_video(v) for v in video])  # [B,F,H,W,C]
            if output_type == "np":
                video = video.cpu().numpy()
        if hasattr(self.vae, "check_range"):
            self.vae.check_range()   # a VAE call of this job left the fp16 range: fail before handing anything back
        if tr.enabled:
            self.timing = tr.summary()
            if getattr(tr, "path", None):
                tr.finish()
        if not return_dict:
            return (video,)
        return WanPipelineOutput(frames=video)

    def _model_input(self, latents, condition, dtype):
        """PIPE:590: cat([latents, condition], 1).to(transformer_dtype)."""
        x = torch.cat([latents.to(torch.float32), condition], dim=1)
        return ops.cast(x, dtype)
