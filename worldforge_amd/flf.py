"""FLF (flow-gated latent fusion) channel selector, HIP-backed.

Mirror of `VideoMotionPCASelector.select_motion_related_channels` (SCHED:338-437) and its metric
`_compute_flow_metrics` (SCHED:497-607).  The per-channel motion extraction + the three-way metric reduction run on the
GPU for all 16 channels in two launches; the 16 similarities come back in ONE device->host copy (the reference does 32
D2H copies and 16 `.item()` syncs per call), and the threshold logic (16 scalars) stays on the host as in the reference.

flow_backend:
  "farneback" -- DEFAULT: what the reference computes as deployed (requirements.txt:8 installs opencv-python and
                 use_optical_flow defaults to True): SCHED:156-248, cv2.calcOpticalFlowFarneback with the parameters of
                 :220-224, as a batched GPU implementation (csrc/flow.hip): all 2 x 16 x (T-1) frame pairs in ~30 small
                 launches, nothing copied to the host.  OpenCV is a third-party dependency absent from /root/reference and
                 from this image, so this backend is checked against oracle/farneback.py (a restatement of the published
                 algorithm) only: PARITY WITH A REAL cv2 IS UNPINNED.
  "tdiff"     -- temporal difference motion (SCHED:391-392 / 478-479): the branch the reference executes only when
                 `import cv2` fails (SCHED:159-161 -> except at :390); it is the branch the golden trajectories were
                 recorded on (no cv2 in the build container), so the golden tests select it explicitly.
"""
from __future__ import annotations

from typing import List

import numpy as np
import torch

from . import ops


class VideoMotionPCASelector:
    def __init__(self, flow_backend: str = "farneback"):
        if flow_backend not in ("tdiff", "farneback"):
            raise ValueError(f"unknown flow_backend {flow_backend!r}")
        self.flow_backend = flow_backend
        self.last_similarities = None

    def channel_similarities(self, pred_original_sample: torch.Tensor, video_latents: torch.Tensor) -> np.ndarray:
        """SCHED:373-397 + 439-495 -> numpy float64 array of C similarities."""
        if pred_original_sample.shape[0] != 1:
            raise NotImplementedError("FLF: batch size 1 only (as the reference's squeeze(0) path)")
        if self.flow_backend == "farneback":
            ref_f = ops.farneback_flows(video_latents[0])          # [C, T-1, 2, h, w] fp32
            ch_f = ops.farneback_flows(pred_original_sample[0])
            sim = ops.flow_metrics(ref_f, ch_f)
            self.last_similarities = sim.cpu().numpy().astype(np.float64)
            return self.last_similarities
        ref_m = ops.temporal_diff(video_latents[0])        # [C, T-1, h, w] fp32 (video_latents promoted to fp32 first)
        ch_m = ops.temporal_diff(pred_original_sample[0])  # [C, T-1, h, w] fp32
        sim = ops.flow_metrics(ref_m.unsqueeze(2), ch_m.unsqueeze(2))
        self.last_similarities = sim.cpu().numpy().astype(np.float64)  # the single sync of the FLF gate
        return self.last_similarities

    @staticmethod
    def select_from_similarities(channel_correlations, current_step: int) -> List[int]:
        """SCHED:408-437."""
        if current_step < 2:
            return []
        corr = np.array(channel_correlations)
        corr_mean, corr_std = np.mean(corr), np.std(corr)
        if current_step <= 10:
            max_replace = 0 if current_step <= 5 else 1
            channels = np.argsort(corr)[:max_replace].tolist()
        else:
            threshold = corr_mean - 0.625 * corr_std
            below = [i for i, s in enumerate(corr) if s < threshold]
            if len(below) < 2:
                channels = np.argsort(corr)[:2].tolist()
            elif len(below) > 6:
                scored = sorted([(i, corr[i]) for i in below], key=lambda x: x[1])
                channels = [i for i, _ in scored[:6]]
            else:
                channels = below
        channels.sort()
        return channels

    def select_motion_related_channels(self, pred_original_sample: torch.Tensor, video_latents: torch.Tensor, mask=None,
                                       keep_channels: int = 12, current_step: int = 0, total_steps: int = 50,
                                       use_optical_flow: bool = True, static: bool = False, **kwargs) -> List[int]:
        """SCHED:338-437."""
        if current_step < 2:
            return []
        if pred_original_sample.dim() != 5 or video_latents.dim() != 5:
            return list(range(min(2, pred_original_sample.shape[1])))
        sims = self.channel_similarities(pred_original_sample, video_latents)
        return self.select_from_similarities(sims, current_step)
