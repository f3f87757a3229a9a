"""FlowMatchEulerDiscreteScheduler of the LongCat-Video path with the WorldForge trajectory injection, HIP-backed.

Host-side mirror of longcat_for_worldforge/longcat_video/modules/scheduling_flow_match_euler_discrete.py (SCHED): same attributes
(`sigmas`, `timesteps`, `_step_index`, `derivative_history`, `resample_sigmas`), same `set_timesteps` / `step` / `add_noise` /
`fuse_latents` signatures for the arguments the guided i2v loop passes (pipeline_longcat_video.py:891-909, 950-955, 978-988).  The
schedule (a few dozen scalars) is computed on the host exactly as SCHED:664-709; every tensor-sized operation runs in libwf_hip.so.

Static-shift configuration only (the released scheduler config): dynamic shifting, karras / exponential / beta sigma conversions,
stochastic sampling and per-token timesteps are not on the WorldForge path and raise.
"""
from __future__ import annotations

from dataclasses import dataclass
from types import SimpleNamespace
from typing import Any, List, Optional

import numpy as np
import torch

from . import ops


@dataclass
class FlowMatchEulerDiscreteSchedulerOutput:
    prev_sample: torch.Tensor
    pred_x0: Optional[torch.Tensor] = None


class VideoMotionChannelSelector:
    """SCHED:35-381.  flow_backend "farneback" (default) = the branch the reference executes where `import cv2` succeeds, i.e. as
    deployed (SCHED:45-51, 286-300: per-channel normalisation, uint8, dense Farneback flow: `wf_farneback_flows` quant_mode 1);
    "tdiff" = temporal-difference motion (SCHED:165-170), the branch taken when the import fails -- OpenCV is absent from
    /root/reference and from this image, so the golden fixtures were recorded through it.  Motion extraction + the three-way metric for
    all 16 channels run in a handful of launches; the 16 similarities come back in ONE device->host copy; the threshold logic (16
    scalars) stays on the host."""

    def __init__(self, flow_backend: str = "farneback"):
        if flow_backend not in ("farneback", "tdiff"):
            raise ValueError(f"flow_backend must be 'farneback' or 'tdiff', got {flow_backend!r}")
        self.flow_backend = flow_backend
        self.use_optical_flow = flow_backend == "farneback"  # SCHED:45-51
        self.last_similarities = None

    def channel_similarities(self, pred_original_sample: torch.Tensor, encoded_video: torch.Tensor) -> np.ndarray:
        if pred_original_sample.shape[0] != 1:
            raise NotImplementedError("FLF: batch size 1 only")
        enc = ops.cast(encoded_video, pred_original_sample.dtype)[0]
        if self.use_optical_flow:
            ref_m = ops.farneback_flows(enc, quant_mode=1)                     # [C, T-1, 2, h, w] fp32
            ch_m = ops.farneback_flows(pred_original_sample[0], quant_mode=1)
            sim = ops.flow_metrics(ref_m, ch_m, variant=1)
        else:
            ref_m = ops.temporal_diff(enc)                                      # [C, T-1, h, w] fp32
            ch_m = ops.temporal_diff(pred_original_sample[0])
            sim = ops.flow_metrics(ref_m.unsqueeze(2), ch_m.unsqueeze(2), variant=1)
        self.last_similarities = sim.cpu().numpy().astype(np.float64)  # the single sync of the FLF gate
        return self.last_similarities

    @staticmethod
    def select_from_similarities(correlations, current_step: int, use_distill: bool = False,
                                 max_replace_threshold: Optional[int] = None) -> List[int]:
        """SCHED:330-381."""
        if current_step < 2:
            return []
        corr = np.array(correlations)
        corr_mean, corr_std = np.mean(corr), np.std(corr)
        if current_step <= (3 if use_distill else 5):
            channels = np.argsort(corr)[:1].tolist()
        else:
            max_replace = max_replace_threshold if max_replace_threshold is not None else (3 if use_distill else 1)
            threshold = corr_mean - 0.625 * corr_std
            below = [i for i, s in enumerate(corr) if s < threshold]
            if len(below) < 1:
                channels = np.argsort(corr)[:1].tolist()
            elif len(below) > max_replace:
                scored = sorted([(i, corr[i]) for i in below], key=lambda x: x[1])
                channels = [i for i, _ in scored[:max_replace]]
            else:
                channels = below
        return sorted(channels)

    def select_motion_related_channels(self, pred_original_sample: torch.Tensor, encoded_video: torch.Tensor, current_step: int = 0,
                                       total_steps: int = 50, static: bool = False, use_distill: bool = False,
                                       max_replace_threshold: Optional[int] = None) -> List[int]:
        """SCHED:245-381."""
        if current_step < 2:
            return []
        if pred_original_sample.dim() != 5 or encoded_video.dim() != 5:
            return []
        sims = self.channel_similarities(pred_original_sample, encoded_video)
        return self.select_from_similarities(sims, current_step, use_distill, max_replace_threshold)


class FlowMatchEulerDiscreteScheduler:
    order = 1

    def __init__(self, num_train_timesteps: int = 1000, shift: float = 1.0, use_dynamic_shifting: bool = False,
                 invert_sigmas: bool = False, shift_terminal: Optional[float] = None, use_karras_sigmas: bool = False,
                 use_exponential_sigmas: bool = False, use_beta_sigmas: bool = False, stochastic_sampling: bool = False, flow_backend: str = "farneback",
                 **unused):
        if use_dynamic_shifting or invert_sigmas or shift_terminal or use_karras_sigmas or use_exponential_sigmas or use_beta_sigmas \
                or stochastic_sampling:
            raise NotImplementedError("only the static-shift deterministic Euler configuration of the WorldForge path is built")
        self.config = SimpleNamespace(num_train_timesteps=num_train_timesteps, shift=shift, use_dynamic_shifting=False,
                                      stochastic_sampling=False)
        # SCHED:473-489
        ts = np.linspace(1, num_train_timesteps, num_train_timesteps, dtype=np.float32)[::-1].copy()
        sig = torch.from_numpy(ts).to(torch.float32) / num_train_timesteps
        sig = shift * sig / (1 + (shift - 1) * sig)
        self.timesteps = sig * num_train_timesteps
        self.sigmas = sig
        self.sigma_min, self.sigma_max = self.sigmas[-1].item(), self.sigmas[0].item()
        self._shift = shift
        self.flow_backend = flow_backend  # FLF motion extraction: see VideoMotionChannelSelector
        self._step_index = None
        self._begin_index = None
        self.num_inference_steps = None
        self.derivative_history: list = []
        self.resample_sigmas = None
        self.resample_timesteps = None
        self.is_resampling = False
        self._channel_selector = None

    @property
    def shift(self):
        return self._shift

    @property
    def step_index(self):
        return self._step_index

    @property
    def begin_index(self):
        return self._begin_index

    def set_begin_index(self, begin_index: int = 0):
        self._begin_index = begin_index

    # ---- SCHED:610-716 -------------------------------------------------------------------------------------------------
    def set_timesteps(self, num_inference_steps: Optional[int] = None, device=None, sigmas=None, mu=None, timesteps=None):
        if timesteps is not None:
            raise NotImplementedError("custom timesteps are not used by the WorldForge path")
        if sigmas is None:
            t = np.linspace(self.sigma_max * self.config.num_train_timesteps, self.sigma_min * self.config.num_train_timesteps,
                            num_inference_steps)
            s = (t / self.config.num_train_timesteps)
        else:
            s = (sigmas.cpu().numpy() if isinstance(sigmas, torch.Tensor) else np.array(sigmas)).astype(np.float32)
            if num_inference_steps is not None and len(s) != num_inference_steps:
                raise ValueError("`sigmas` and `timesteps` should have the same length as num_inference_steps, if `num_inference_steps` is provided")
        self.num_inference_steps = len(s)
        s = self.shift * s / (1 + (self.shift - 1) * s)
        sig = torch.from_numpy(np.asarray(s)).to(dtype=torch.float32)
        self.timesteps = sig * self.config.num_train_timesteps  # host tensors: a few dozen scalars
        self.sigmas = torch.cat([sig, torch.zeros(1)])
        self._step_index = None
        self._begin_index = None
        self._compute_resample_sigmas_and_timesteps()
        self.derivative_history = []

    # ---- SCHED:1005-1039 -----------------------------------------------------------------------------------------------
    def _compute_resample_sigmas_and_timesteps(self):
        if len(self.sigmas) < 2:
            self.resample_sigmas = self.resample_timesteps = None
            return
        self.resample_sigmas = self.sigmas[:-1].clone()
        self.resample_timesteps = self.resample_sigmas * self.config.num_train_timesteps

    def set_resample_mode(self, enabled: bool):
        self.is_resampling = enabled

    def get_resample_timestep(self, step_index: int) -> torch.Tensor:
        if self.resample_timesteps is not None and step_index < len(self.resample_timesteps):
            return self.resample_timesteps[step_index]
        return self.timesteps[min(step_index, len(self.timesteps) - 1)]

    # ---- SCHED:718-738 -------------------------------------------------------------------------------------------------
    def index_for_timestep(self, timestep, schedule_timesteps=None):
        if schedule_timesteps is None:
            schedule_timesteps = self.timesteps
        t = torch.as_tensor(timestep).detach().to("cpu", torch.float32)
        indices = (schedule_timesteps.cpu() == t).nonzero()
        pos = 1 if len(indices) > 1 else 0
        return indices[pos].item()

    def _init_step_index(self, timestep):
        self._step_index = self.index_for_timestep(timestep) if self.begin_index is None else self._begin_index

    # ---- SCHED:740-912 -------------------------------------------------------------------------------------------------
    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor, s_churn: float = 0.0, s_tmin: float = 0.0,
             s_tmax: float = float("inf"), s_noise: float = 1.0, generator=None, per_token_timesteps=None, return_dict: bool = True,
             video_ref: Optional[torch.Tensor] = None, mask: Optional[torch.Tensor] = None, guided: bool = False,
             resampling: bool = False, vae: Optional[Any] = None, use_pca_channel_selection: bool = False, static: bool = False,
             current_step: int = -1, total_steps: int = 50, sample_full: Optional[torch.Tensor] = None, use_distill: bool = False,
             max_replace_threshold: Optional[int] = None):
        if isinstance(timestep, int) or (isinstance(timestep, torch.Tensor) and timestep.dtype in (torch.int32, torch.int64)):
            raise ValueError("Passing integer indices (e.g. from `enumerate(timesteps)`) as timesteps to"
                             " `FlowMatchEulerDiscreteScheduler.step()` is not supported. Make sure to pass"
                             " one of the `scheduler.timesteps` as a timestep.")
        if per_token_timesteps is not None:
            raise NotImplementedError("per-token timesteps are not used by the WorldForge path")
        if self.step_index is None:
            self._init_step_index(timestep)
        sample = ops.cast(sample, torch.float32)
        sigma = np.float32(self.sigmas[self.step_index].item())
        dt = np.float32(self.sigmas[self.step_index + 1].item()) - sigma
        pred_x0 = ops.x0_from_v(sample, model_output, float(sigma))  # SCHED:836
        if guided and video_ref is not None and not resampling and sample_full is not None:
            full_v = torch.cat([torch.zeros_like(model_output[:, :, 0:1]), model_output], dim=2)  # first frame: no noise prediction
            x0_full = ops.x0_from_v(ops.cast(sample_full, torch.float32), full_v, float(sigma))
            fused = self.fuse_latents(pred_original_sample=x0_full, video_latents=video_ref, mask=mask, vae=vae,
                                      use_pca_channel_selection=use_pca_channel_selection, static=static, current_step=current_step,
                                      total_steps=total_steps, max_replace_threshold=max_replace_threshold, use_distill=use_distill)
            pred_x0 = fused[:, :, 1:, :, :]
        self.derivative_history.append(model_output)
        prev_sample = ops.x0_from_v(sample, model_output, float(-dt))  # SCHED:894: sample + dt * model_output
        self._step_index += 1
        prev_sample = ops.cast(prev_sample, model_output.dtype)
        if not return_dict:
            return (prev_sample,)
        return FlowMatchEulerDiscreteSchedulerOutput(prev_sample=prev_sample, pred_x0=pred_x0)

    # ---- SCHED:1041-1070 -----------------------------------------------------------------------------------------------
    def add_noise(self, original_samples: torch.Tensor, noise: torch.Tensor, timesteps: torch.Tensor,
                  use_resample_sigma: bool = False) -> torch.Tensor:
        ts = torch.as_tensor(timesteps).reshape(-1)
        idx = [self.index_for_timestep(t) for t in ts]
        if len(set(idx)) != 1:
            raise NotImplementedError("one timestep per call (batch size 1, as the reference path)")
        sigma = np.float32(self.sigmas[idx[0]].item())
        return ops.add_noise(original_samples, noise, float(np.float32(1.0) - sigma), float(sigma))

    # ---- SCHED:1072-1233 -----------------------------------------------------------------------------------------------
    def fuse_latents(self, pred_original_sample: torch.Tensor, video_latents: torch.Tensor, mask: torch.Tensor, vae: Any,
                     use_pca_channel_selection: bool = False, static: bool = False, current_step: int = 0, total_steps: int = 50,
                     use_distill: bool = False, max_replace_threshold: Optional[int] = None) -> torch.Tensor:
        """De-normalise -> VAE decode -> reference pixels where the mask is set -> VAE encode (mode) -> re-normalise -> FLF.  The
        reference and the mask must already have the decoded size (no alignment on this path); on a mismatch the reference logs the
        ValueError raised inside its try block and returns the prediction unchanged -- mirrored."""
        if mask is None or video_latents is None or vae is None:
            return pred_original_sample
        x0 = pred_original_sample
        mean, std = vae.config.latents_mean, vae.config.latents_std
        if getattr(vae, "comm", None) is not None and hasattr(vae, "decode_blend_encode") and x0.dim() == 5 and x0.shape[0] == 1 \
                and vae.can_shard(x0.shape[3]):
            # row-sharded VAE: the decoded video is never gathered (vae.decode_blend_encode); the decoded size follows from the latent's
            tds, sds = 2 ** sum(vae.temperal_downsample), 2 ** len(vae.temperal_downsample)
            dshape = (1, 3, (x0.shape[2] - 1) * tds + 1, x0.shape[3] * sds, x0.shape[4] * sds)
            if tuple(video_latents.shape) != dshape or mask.shape[1] != 1 or tuple(mask.shape[2:]) != dshape[2:]:
                return pred_original_sample
            enc = vae.decode_blend_encode(self._to_vae(ops.latent_denorm(x0, mean, std), vae), video_latents, mask).mode()
        else:
            cols = {}
            if hasattr(vae, "needed_columns") and getattr(vae, "crop_to_mask", False) and x0.dim() == 5 and mask.dim() == 5 \
                    and mask.shape[1] == 1 and mask.dtype == torch.float32 and mask.is_contiguous():
                cols = {"columns": vae.needed_columns(mask)}   # only what the blend can see is decoded (vae.decode(columns=...))
            decoded = vae.decode(self._to_vae(ops.latent_denorm(x0, mean, std), vae), return_dict=False, **cols)[0]   # SCHED:1124 + 1127
            if tuple(video_latents.shape) != tuple(decoded.shape) or mask.shape[1] != 1 or tuple(mask.shape[2:]) != tuple(decoded.shape[2:]):
                return pred_original_sample
            fused = ops.blend_pixels(video_latents, mask, decoded)   # in the decoded video's dtype (SCHED:1152-1164)
            enc = vae.encode(self._to_vae(fused, vae)).latent_dist.mode()   # SCHED:1166 + 1169
        if tuple(enc.shape) != tuple(x0.shape):
            return pred_original_sample
        # SCHED:1183: the constants are in the prediction's dtype (fp32), which promotes a bf16 module's latents
        enc = ops.latent_norm(ops.cast(enc, torch.float32), mean, std, const_dtype=x0.dtype)
        if use_pca_channel_selection:
            if self._channel_selector is None:
                self._channel_selector = VideoMotionChannelSelector(self.flow_backend)
            channels = self._channel_selector.select_motion_related_channels(
                pred_original_sample=x0, encoded_video=enc, current_step=current_step, total_steps=total_steps, static=static,
                use_distill=use_distill, max_replace_threshold=max_replace_threshold)
            ops.channel_swap_(enc, x0, channels)
        if hasattr(vae, "check_range"):   # fp16 range flag of this round trip (see scheduler.UniPCMultistepScheduler.fuse_latents)
            vae.check_range(wait=bool(use_pca_channel_selection))
        return ops.cast(enc, x0.dtype)

    @staticmethod
    def _to_vae(x: torch.Tensor, vae) -> torch.Tensor:
        """`.to(dtype=vae.dtype)` (SCHED:1124, 1166): the LongCat entry loads its VAE in bf16 (run_longcat_worldforge_single.py:205)."""
        vdt = getattr(vae, "dtype", torch.float32)
        return x if x.dtype == vdt else ops.cast(x, vdt)

    def __len__(self):
        return self.config.num_train_timesteps
