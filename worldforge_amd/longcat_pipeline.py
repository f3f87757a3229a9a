"""The LongCat-Video guided image-to-video sampler (IRR re-noising, FLF gate, DSG auto-guidance, CFG-zero), HIP-backed.

Host-side mirror of `LongCatVideoPipeline.generate_i2v` (PIPE = longcat_for_worldforge/longcat_video/pipeline_longcat_video.py:619-1006)
and of `generate_refine` (PIPE:1271-1511, the 720p refine pass):
same sampling knobs, same control flow (PIPE:823-994), same RNG draw order (CPU generator: noise latents PIPE:256, the posterior sample
of the conditioning frame PIPE:278, the re-noise draws PIPE:925), same dtype hand-offs (fp32 latents, DiT input / timesteps in the
DiT dtype).  The DiT, the VAE and the scheduler are objects speaking the reference's call protocol, so the MI355X-native modules of
this package (longcat_dit.LongCatVideoTransformer3DModel, vae.AutoencoderKLWan, longcat_scheduler.FlowMatchEulerDiscreteScheduler)
and test doubles are interchangeable.  The UMT5 text encoder runs once per video outside the loop and is out of scope: its outputs
are taken as tensors (`prompt_embeds` [1,1,N,C] + `prompt_attention_mask` [1,N], and the negative pair).  The target size is given
explicitly (`height`, `width`): the resolution-bucket lookup of PIPE:358-372 belongs to the front-end.
"""
from __future__ import annotations

from typing import Optional, Union

import numpy as np
import torch

from . import ops


class LongCatVideoPipeline:
    def __init__(self, vae, scheduler, dit, device: Union[str, torch.device] = "cuda:0"):
        self.vae, self.scheduler, self.dit = vae, scheduler, dit
        self.device = torch.device(device)
        cfg = getattr(vae, "config", None)
        self.vae_scale_factor_temporal = getattr(cfg, "scale_factor_temporal", 4)  # PIPE:83-84
        self.vae_scale_factor_spatial = getattr(cfg, "scale_factor_spatial", 8)
        self._num_timesteps = 1000
        self._num_distill_sample_steps = 50
        self._guidance_scale = 1.0

    @property
    def guidance_scale(self):
        return self._guidance_scale

    @property
    def do_classifier_free_guidance(self):
        return self._guidance_scale > 1.0

    # ---- PIPE:317-331 ---------------------------------------------------------------------------------------------------
    def get_timesteps_sigmas(self, sampling_steps: int, use_distill: bool = False) -> torch.Tensor:
        if use_distill:
            idx = torch.arange(1, self._num_distill_sample_steps + 1, dtype=torch.float32)
            idx = (idx * (self._num_timesteps // self._num_distill_sample_steps)).round().long()
            inf = np.floor(np.linspace(0, self._num_distill_sample_steps, num=sampling_steps, endpoint=False)).astype(np.int64)
            sigmas = torch.flip(idx, [0])[inf].float() / self._num_timesteps
            sigmas = sigmas - sigmas[-1]
        else:
            sigmas = torch.linspace(0.999, 0.000, sampling_steps)
        return sigmas.to(torch.float32)

    def _preprocess_image(self, image, height, width) -> torch.Tensor:
        """diffusers VideoProcessor.preprocess: -> [1,3,H,W] fp32 in [-1,1]."""
        if isinstance(image, torch.Tensor):
            t = (image if image.dim() == 4 else image.unsqueeze(0)).to(torch.float32)
        else:
            if image.size != (width, height):
                image = image.resize((width, height))
            t = torch.from_numpy(np.array(image).astype(np.float32) / 255.0).permute(2, 0, 1).unsqueeze(0)
        if t.shape[-2:] != (height, width):
            raise ValueError(f"image tensor is {tuple(t.shape[-2:])}, expected {(height, width)}")
        return 2.0 * t - 1.0

    def _randn(self, shape, generator):
        # diffusers / the reference draw on the CPU generator and move (RNG parity)
        if generator is not None and generator.device.type == "cpu":
            return torch.randn(shape, generator=generator).to(self.device)
        return torch.randn(shape, generator=generator, device=self.device)

    # ---- PIPE:214-286 with num_cond_frames = 1 ----------------------------------------------------------------------------
    def prepare_latents(self, image: torch.Tensor, batch_size: int, num_channels_latents: int, height: int, width: int,
                        num_frames: int, generator=None, latents: Optional[torch.Tensor] = None) -> torch.Tensor:
        T = (num_frames - 1) // self.vae_scale_factor_temporal + 1
        shape = (batch_size, num_channels_latents, T, height // self.vae_scale_factor_spatial, width // self.vae_scale_factor_spatial)
        latents = self._randn(shape, generator).to(torch.float32) if latents is None else latents.to(self.device, torch.float32)
        cond = []
        for i in range(batch_size):
            post = self.vae.encode(image[i].unsqueeze(0).unsqueeze(2)).latent_dist
            cond.append(post.sample(generator))  # PIPE:278 retrieve_latents(..., sample_mode="sample")
        cond = torch.cat(cond, dim=0).to(self.device, torch.float32)
        latents[:, :, :1] = ops.latent_norm(cond, self.vae.config.latents_mean, self.vae.config.latents_std)
        return latents

    # ---- PIPE:1271-1511 ---------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def generate_refine(self, stage1_video, height: int, width: int, prompt_embeds: torch.Tensor, prompt_attention_mask: torch.Tensor,
                        image=None, num_cond_frames: int = 0, num_inference_steps: int = 50, generator=None, output_type: str = "np",
                        t_thresh: float = 0.5, spatial_refine_only: bool = False, step_hook=None):
        """The 720p refine pass: the stage-1 (480p) video is up-sampled, encoded, mixed with noise at t_thresh and denoised from there
        without CFG, the DiT running with block-sparse self-attention (enable it on the DiT, with the refinement LoRA folded in, as
        run_longcat_worldforge_single.py:447-451 does).  stage1_video: uint8 frames [F, H0, W0, 3] (tensor / array / list of arrays);
        image: the conditioning first frame at the target size or None.  Returns frames [1, F', H, W, 3] in [0, 1]."""
        import math

        from ._ffi import call
        dev, sch = self.device, self.scheduler
        ssp = self.vae_scale_factor_spatial * 2 * 4  # PIPE:1336
        if height % ssp != 0 or width % ssp != 0:
            raise ValueError(f"`height and width` have to be divisible by {ssp} but are {height} and {width}.")
        dit_dtype = self.dit.dtype
        pe, pm = prompt_embeds.to(dev, dit_dtype), prompt_attention_mask.to(dev)
        # PIPE:1394-1402: the schedule, cut at t_thresh
        sch.set_timesteps(num_inference_steps, sigmas=self.get_timesteps_sigmas(num_inference_steps), device=dev)
        timesteps = sch.timesteps
        if t_thresh:
            tt = torch.tensor(t_thresh * 1000, dtype=timesteps.dtype)
            timesteps = torch.cat([tt.unsqueeze(0), timesteps[timesteps < tt]])
            sch.timesteps = timesteps
            sch.sigmas = torch.cat([timesteps / 1000, torch.zeros(1)])
        # PIPE:1404-1413: up-sampling chain in the DiT dtype, one kernel
        frames = torch.as_tensor(np.array(stage1_video)) if not isinstance(stage1_video, torch.Tensor) else stage1_video
        if frames.dtype != torch.uint8 or frames.dim() != 4 or frames.shape[-1] != 3:
            raise ValueError("stage1_video must be uint8 frames [F, H, W, 3]")
        if dit_dtype != torch.bfloat16:
            raise NotImplementedError("the up-sampling kernel rounds as a bf16 model does")
        frames = frames.to(dev).contiguous()
        nf, H0, W0, _ = frames.shape
        new_frame_size = nf if spatial_refine_only else 2 * nf
        up = torch.empty((1, 3, new_frame_size, height, width), dtype=torch.float32, device=dev)
        call("wf_refine_upsample_u8", frames.data_ptr(), up.data_ptr(), nf, H0, W0, new_frame_size, height, width, ops.stream())
        # PIPE:1415-1435: pad to the block-sparse granularity (4 latent frames), encode, mix with noise
        gran, tsc = 4, self.vae_scale_factor_temporal
        num_noise_frames = new_frame_size - num_cond_frames
        ncl = added_c = 0
        if num_cond_frames > 0:
            ncl = 1 + math.ceil((num_cond_frames - 1) / tsc)
            ncl = math.ceil(ncl / gran) * gran
            added_c = 1 + (ncl - 1) * tsc - num_cond_frames
            num_cond_frames = num_cond_frames + added_c
        nnl = math.ceil(math.ceil(num_noise_frames / tsc) / gran) * gran
        added_n = nnl * tsc - num_noise_frames
        up = torch.cat([up[:, :, 0:1].repeat(1, 1, added_c, 1, 1), up, up[:, :, -1:].repeat(1, 1, added_n, 1, 1)], dim=2)
        mean, std = self.vae.config.latents_mean, self.vae.config.latents_std
        samp = self.vae.encode(up if getattr(self.vae, "dtype", torch.float32) == torch.float32 else ops.cast(up, self.vae.dtype)) \
            .latent_dist.sample(generator).to(dev)
        if samp.dtype == torch.float32:
            lat = ops.latent_norm(samp, mean, std)
            noise = self._randn(tuple(lat.shape), generator).to(lat.dtype)
            latents = ops.add_noise(lat, noise, 1 - t_thresh, t_thresh)  # (1 - t) * latent + t * noise
        else:
            # a bf16 VAE module hands back a bf16 sample: PIPE:1431-1433 then normalise in bf16 (constants in bf16), draw the noise IN BF16
            # (a different stream of the generator than an fp32 draw) and mix in bf16; prepare_latents casts the result to fp32 (PIPE:234)
            m = torch.tensor(mean).view(1, -1, 1, 1, 1).to(dev, samp.dtype)
            istd = 1.0 / torch.tensor(std).view(1, -1, 1, 1, 1).to(dev, samp.dtype)
            lat = (samp - m) * istd
            if generator is not None and generator.device.type == "cpu":
                noise = torch.randn(lat.shape, generator=generator, dtype=lat.dtype).to(dev)
            else:
                noise = torch.randn(lat.shape, generator=generator, dtype=lat.dtype, device=dev)
            latents = ((1 - t_thresh) * lat + t_thresh * noise).to(torch.float32)
        del up, noise
        if image is not None:  # PIPE:262-284: the condition frame, front-padded, encoded (posterior sample), normalised
            img = self._preprocess_image(image, height, width).to(dev, dit_dtype)
            enc_in = img[0].unsqueeze(0).unsqueeze(2)
            enc_in = torch.cat([enc_in[:, :, 0:1].repeat(1, 1, added_c, 1, 1), enc_in], dim=2)
            assert enc_in.shape[2] == num_cond_frames
            cond = self.vae.encode(enc_in).latent_dist.sample(generator).to(dev, torch.float32)
            latents[:, :, :ncl] = ops.latent_norm(cond, mean, std)
        elif num_cond_frames > 0:
            raise ValueError("num_cond_frames > 0 needs the conditioning image (video conditioning is not built)")
        # PIPE:1464-1497
        for i, t in enumerate(timesteps):
            if step_hook is not None:
                step_hook(i, "start")
            ts = t.expand(latents.shape[0]).to(dit_dtype).unsqueeze(-1).repeat(1, latents.shape[2])
            ts[:, :ncl] = 0
            noise_pred = -self.dit(hidden_states=ops.cast(latents, dit_dtype), timestep=ts, encoder_hidden_states=pe,
                                   encoder_attention_mask=pm, num_cond_latents=ncl)
            latents[:, :, ncl:] = sch.step(noise_pred[:, :, ncl:], t, latents[:, :, ncl:], return_dict=False)[0]
            if step_hook is not None:
                step_hook(i, "end")
        if output_type == "latent":
            self._check_vae_range()
            return latents
        video = self.vae.decode(self._final_latents(latents), return_dict=False)[0]
        video = torch.stack([ops.postprocess_video(v) for v in video])[:, added_c: new_frame_size + added_c]  # PIPE:1505
        self._check_vae_range()
        return video.cpu().numpy() if output_type == "np" else video

    def _final_latents(self, latents: torch.Tensor) -> torch.Tensor:
        """PIPE:999-1000: `latents.to(self.vae.dtype)` and the de-normalisation IN THAT DTYPE (a bf16 VAE module, the LongCat entry's
        run_longcat_worldforge_single.py:205: constants and both statements in bf16), handed to decode in the module dtype."""
        vdt = getattr(self.vae, "dtype", torch.float32)
        z = ops.latent_denorm(ops.cast(latents, vdt), self.vae.config.latents_mean, self.vae.config.latents_std)  # f32 holding vdt values
        return z if vdt == torch.float32 else ops.cast(z, vdt)

    def _check_vae_range(self):
        """A VAE call of this job left the fp16 operand range (vae.AutoencoderKLWan.check_range): fail before handing anything back."""
        if hasattr(self.vae, "check_range"):
            self.vae.check_range()

    # ---- PIPE:619-1006 ----------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def generate_i2v(self, image, height: int, width: int, prompt_embeds: torch.Tensor, prompt_attention_mask: torch.Tensor,
                     negative_prompt_embeds: Optional[torch.Tensor] = None, negative_prompt_attention_mask: Optional[torch.Tensor] = None,
                     num_frames: int = 93, num_inference_steps: int = 50, use_distill: bool = False, guidance_scale: float = 4.0,
                     generator=None, latents: Optional[torch.Tensor] = None, output_type: str = "np",
                     video_ref: Optional[torch.Tensor] = None, mask: Optional[torch.Tensor] = None, guided: bool = False,
                     resample_steps: int = 3, guide_steps: int = 20, resample_round: int = 20, omega: float = 1.8,
                     omega_resample: float = 1.0, use_pca_channel_selection: bool = False, static: bool = False,
                     max_replace_threshold: Optional[int] = None, step_hook=None):
        dev, sch = self.device, self.scheduler
        ssp = self.vae_scale_factor_spatial * 2
        if height % ssp != 0 or width % ssp != 0:
            raise ValueError(f"`height and width` have to be divisible by {ssp} but are {height} and {width}.")  # PIPE:199-201
        if num_frames % self.vae_scale_factor_temporal != 1:
            num_frames = num_frames // self.vae_scale_factor_temporal * self.vae_scale_factor_temporal + 1  # PIPE:700-704
        num_frames = max(num_frames, 1)
        self._guidance_scale = guidance_scale
        do_cfg = self.do_classifier_free_guidance
        dit_dtype = self.dit.dtype
        pe = prompt_embeds.to(dev, dit_dtype)
        pm = prompt_attention_mask.to(dev)
        if do_cfg:
            if negative_prompt_embeds is None:
                raise ValueError("classifier-free guidance (guidance_scale > 1) needs the negative prompt embeddings")
            pe = torch.cat([negative_prompt_embeds.to(dev, dit_dtype), pe], dim=0)  # PIPE:760-762
            pm = torch.cat([negative_prompt_attention_mask.to(dev), pm], dim=0)
        # PIPE:764-767
        sch.set_timesteps(num_inference_steps, sigmas=self.get_timesteps_sigmas(num_inference_steps, use_distill=use_distill), device=dev)
        timesteps = sch.timesteps
        # PIPE:769-789
        img = self._preprocess_image(image, height, width).to(dev, dit_dtype)
        latents = self.prepare_latents(img, 1, self.dit.config.in_channels, height, width, num_frames, generator, latents)
        if video_ref is not None and guided:
            video_ref = torch.as_tensor(video_ref).to(dev, torch.float32)  # PIPE:792-801
        if mask is not None and guided:  # PIPE:803-817
            mask = torch.from_numpy(mask) if isinstance(mask, np.ndarray) else mask
            if mask.dim() == 3:
                mask = mask.unsqueeze(0).unsqueeze(1)
            elif mask.dim() == 4 and mask.shape[1] != 1:
                mask = mask[:, 0:1, :, :].unsqueeze(0)
            elif mask.dim() == 4 and mask.shape[0] == 1:
                mask = mask.unsqueeze(1)
            elif mask.dim() == 5 and mask.shape[1] != 1:
                mask = mask[:, 0:1, :, :, :]
            mask = mask.to(dev, torch.float32)
        sch.derivative_history = []

        for i, t in enumerate(timesteps):
            if step_hook is not None:
                step_hook(i, "start")
            sch.derivative_history = []
            pred_x0 = None
            scheduler_output = None
            for r in range(resample_steps if (guided and i < resample_round) else 1):
                if r > 0:
                    sch.set_resample_mode(True)
                    sch._step_index -= 1
                else:
                    sch.set_resample_mode(False)
                # PIPE:852-865: the timestep in the DiT dtype, one per latent frame, the condition frame at t = 0
                ts = t.expand(latents.shape[0]).to(dit_dtype)
                x_in = ops.cast(latents, dit_dtype)
                if do_cfg:
                    x_in = torch.cat([x_in] * 2)
                    ts = torch.cat([ts] * 2)
                ts = ts.unsqueeze(-1).repeat(1, x_in.shape[2])
                ts[:, :1] = 0
                split = getattr(self, "cfg_split", None) if do_cfg and x_in.shape[0] == 2 else None
                if split is not None:
                    # CFG groups x sequence shards (parallel.Comm.split, as pipeline.WanImageToVideoPipeline.cfg_split): the CFG
                    # batch [negative, positive] of pipeline_longcat_video.py:857-866 is two samples -- group b of the job's ranks runs sample
                    # b as ONE forward, one all-gather over the whole job hands every rank both velocities (slots 0 and P / 2)
                    world_comm, b = split
                    own = self.dit(hidden_states=x_in[b:b + 1], timestep=ts[b:b + 1], encoder_hidden_states=pe[b:b + 1],
                                   encoder_attention_mask=pm[b:b + 1], num_cond_latents=1).contiguous()
                    both = torch.empty((world_comm.world,) + tuple(own.shape), dtype=own.dtype, device=own.device)
                    world_comm.all_gather(both, own)
                    noise_pred = torch.cat([both[0], both[world_comm.world // 2]])
                else:
                    noise_pred = self.dit(hidden_states=x_in, timestep=ts, encoder_hidden_states=pe, encoder_attention_mask=pm,
                                          num_cond_latents=1)
                if do_cfg:
                    # CFG-zero (PIPE:875-885) and the sign flip of PIPE:888 in one launch per sample
                    u, c = noise_pred.chunk(2)
                    noise_pred = torch.stack([ops.cfg_zero(c[b], u[b], guidance_scale, negate=True) for b in range(c.shape[0])])
                else:
                    noise_pred = -noise_pred
                scheduler_output = sch.step(
                    noise_pred[:, :, 1:], t, latents[:, :, 1:], video_ref=video_ref, mask=mask, guided=guided and i < guide_steps,
                    resampling=r > 0, vae=self.vae, use_pca_channel_selection=use_pca_channel_selection, static=static,
                    current_step=i, total_steps=len(timesteps), sample_full=latents, use_distill=use_distill,
                    max_replace_threshold=max_replace_threshold, return_dict=True)
                if scheduler_output.pred_x0 is not None:
                    pred_x0 = scheduler_output.pred_x0
                if i >= resample_round:
                    break
                if r < resample_steps - 1 and pred_x0 is not None:
                    noise = self._randn(tuple(pred_x0.shape), generator).to(pred_x0.dtype)
                    latents[:, :, 1:] = sch.add_noise(pred_x0, noise, t.expand(pred_x0.shape[0]), use_resample_sigma=False)
            sch.set_resample_mode(False)
            if i < resample_round and len(sch.derivative_history) > 1 and guided:
                # DSG (PIPE:945-976): cosine-corrected extrapolation from the first to the last prediction of this step
                current_omega = omega_resample if i >= guide_steps else omega
                better = ops.dsg(sch.derivative_history[-1].contiguous(), sch.derivative_history[0].contiguous(), current_omega)
                sch._step_index -= 1
                out = sch.step(better, t, latents[:, :, 1:], guided=False, resampling=False, vae=self.vae, sample_full=latents,
                               use_distill=use_distill, return_dict=True)
                latents[:, :, 1:] = out.prev_sample
            elif scheduler_output is not None:
                latents[:, :, 1:] = scheduler_output.prev_sample
            if step_hook is not None:
                step_hook(i, "end")

        if output_type == "latent":
            self._check_vae_range()
            return latents
        video = self.vae.decode(self._final_latents(latents), return_dict=False)[0]
        video = torch.stack([ops.postprocess_video(v) for v in video])  # [B,F,H,W,C] in [0,1]
        self._check_vae_range()
        return video.cpu().numpy() if output_type == "np" else video
