"""Oracle: per-step trajectory injection (IRR pixel blend, FLF channel gate, DSG) -- torch CPU.  TEST INFRASTRUCTURE ONLY.

Restates SCHED = utils/scheduling_unipc_multistep_clean.py and PIPE = utils/pipeline_wan_i2v_clean.py of the reference.
"""
from __future__ import annotations

from typing import Callable, List, Sequence

import numpy as np
import torch
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------------------------
# IRR injection (SCHED:1248-1421)
# ------------------------------------------------------------------------------------------------------------
def latent_denorm(z: torch.Tensor, mean: Sequence[float], std: Sequence[float]) -> torch.Tensor:
    """SCHED:1272-1282: z / (1/std) + mean, result cast to fp32."""
    c = z.shape[1]
    m = torch.tensor(mean).view(1, c, 1, 1, 1).to(z.dtype)
    istd = 1.0 / torch.tensor(std).view(1, c, 1, 1, 1).to(z.dtype)
    return (z / istd + m).to(torch.float32)


def latent_norm(mu: torch.Tensor, mean: Sequence[float], std: Sequence[float], dtype) -> torch.Tensor:
    """SCHED:1385 / PIPE:351: (mu - mean) * (1/std), constants held in `dtype`."""
    c = mu.shape[1]
    m = torch.tensor(mean).view(1, c, 1, 1, 1).to(dtype)
    istd = 1.0 / torch.tensor(std).view(1, c, 1, 1, 1).to(dtype)
    return (mu - m) * istd


def align_reference(ref: torch.Tensor, mask: torch.Tensor, target_shape) -> tuple:
    """SCHED:1300-1371.  Spatial bilinear (align_corners=False) for the reference video, nearest for the mask.
    A frame-count mismatch reaches F.interpolate(size=(T,H,W)) on a 4-D tensor in the reference (SCHED:1326-1334,
    1364-1371), which raises ValueError; the oracle raises the same error class."""
    B, C, Fr, H, W = target_shape
    if ref.shape != tuple(target_shape):
        if ref.shape[0] != B:
            ref = ref.repeat(B, 1, 1, 1, 1)
        b, c, f, h, w = ref.shape
        if (h, w) != (H, W):
            r = F.interpolate(ref.reshape(b * c * f, h, w).unsqueeze(1), size=(H, W), mode="bilinear", align_corners=False)
            ref = r.squeeze(1).reshape(b, c, f, H, W)
        if f != Fr:
            raise ValueError(f"reference video has {f} frames but the decoded video has {Fr}")
    tm = (B, 1, Fr, H, W)
    if mask.shape != tm:
        if mask.shape[0] != B:
            mask = mask.repeat(B, 1, 1, 1, 1)
        if mask.shape[1] != 1:
            mask = mask[:, 0:1]
        b, c, f, h, w = mask.shape
        if (h, w) != (H, W):
            m = F.interpolate(mask.reshape(b * c * f, h, w).unsqueeze(1), size=(H, W), mode="nearest")
            mask = m.squeeze(1).reshape(b, c, f, H, W)
        if f != Fr:
            raise ValueError(f"mask has {f} frames but the decoded video has {Fr}")
    return ref, mask


def blend_pixels(ref: torch.Tensor, mask: torch.Tensor, dec: torch.Tensor) -> torch.Tensor:
    """SCHED:1373-1381."""
    ref = ref.to(dec.dtype)
    mask = mask.to(dec.dtype)
    ref = 2.0 * ref - 1.0
    if mask.shape[1] != dec.shape[1]:
        mask = mask.repeat(1, dec.shape[1], 1, 1, 1)
    return (ref * mask + dec * (1 - mask)).to(torch.float32)


def fuse_latents(x0: torch.Tensor, ref: torch.Tensor, mask: torch.Tensor, *, decode: Callable, encode_mode: Callable,
                 mean, std, use_flf: bool = False, resampling: bool = False, current_step: int = 0,
                 flow_backend: str = "tdiff") -> torch.Tensor:
    """SCHED:1248-1421.  decode: z(fp32) -> video; encode_mode: video -> posterior mode (16 ch)."""
    dec = decode(latent_denorm(x0, mean, std))
    ref_a, mask_a = align_reference(ref, mask, dec.shape)
    fused = blend_pixels(ref_a, mask_a, dec)
    enc = latent_norm(encode_mode(fused), mean, std, x0.dtype)
    if use_flf and not resampling:
        enc_al = enc.to(x0.dtype)
        chans = select_motion_related_channels(x0, enc_al, current_step, flow_backend=flow_backend)
        for c in chans:
            if 0 <= c < enc.shape[1]:
                enc[:, c] = x0[:, c]
    return enc.to(x0.dtype)


# ------------------------------------------------------------------------------------------------------------
# FLF (SCHED:338-607)
# ------------------------------------------------------------------------------------------------------------
def temporal_diff_motion(ch: torch.Tensor) -> torch.Tensor:
    """SCHED:391-392 / 478-479: [1,1,T,h,w] -> [1,T-1,1,h,w]."""
    return (ch[:, :, 1:] - ch[:, :, :-1]).permute(0, 2, 1, 3, 4)


def flow_similarity(ref_motion: torch.Tensor, chan_motion: torch.Tensor) -> float:
    """SCHED:497-607 with mask=None: M-EPE / M-AE / Fl-all -> similarity in [0,1]."""
    n = min(ref_motion.shape[1], chan_motion.shape[1])
    if n <= 0:
        return 0.0
    r, c = ref_motion[:, :n], chan_motion[:, :n].to(ref_motion.dtype)
    r = r[:, :, :2] if r.shape[2] >= 2 else r.repeat(1, 1, 2, 1, 1)[:, :, :2]
    c = c[:, :, :2] if c.shape[2] >= 2 else c.repeat(1, 1, 2, 1, 1)[:, :, :2]
    d = r - c
    epe = torch.sqrt((d ** 2).sum(dim=2) + 1e-8)
    dot = (r * c).sum(dim=2)
    rn = torch.sqrt((r ** 2).sum(dim=2) + 1e-8)
    cn = torch.sqrt((c ** 2).sum(dim=2) + 1e-8)
    cosang = torch.clamp(dot / (rn * cn + 1e-8), -1.0, 1.0)
    ae = torch.acos(cosang) * 180.0 / torch.pi
    outl = (epe > 3.0) & (epe > rn * 0.05)
    m_epe, m_ae, fl = epe.mean(), ae.mean(), outl.float().mean()
    werr = 0.45 * torch.clamp(m_epe / 10.0, 0.0, 1.0) + 0.45 * torch.clamp(fl / 0.5, 0.0, 1.0) + 0.1 * torch.clamp(
        m_ae / 30.0, 0.0, 1.0)
    return torch.clamp(1.0 - werr, 0.0, 1.0).item()


def channel_similarities(pred: torch.Tensor, enc: torch.Tensor, flow_backend: str = "tdiff") -> List[float]:
    """SCHED:373-397 + 439-495.  flow_backend 'tdiff' = the branch the reference executes when cv2 is absent."""
    enc32 = enc.to(torch.float32)
    sims = []
    if flow_backend == "farneback":
        # SCHED:376-389, 462-476 with cv2 present; the flow itself is oracle/farneback.py (parity with cv2 UNPINNED)
        from . import farneback as fb
        pred32 = pred.to(torch.float32)
        emin, erange = enc32.min(), enc32.max() - enc32.min() + 1e-8
        pmin, prange = pred32.min(), pred32.max() - pred32.min() + 1e-8
        for c in range(pred.shape[1]):
            ref_m = torch.from_numpy(fb.channel_flow(enc32[0, c].numpy(), emin.numpy(), erange.numpy())).unsqueeze(0)
            ch_m = torch.from_numpy(fb.channel_flow(pred32[0, c].numpy(), pmin.numpy(), prange.numpy())).unsqueeze(0)
            sims.append(flow_similarity(ref_m, ch_m))
        return sims
    if flow_backend != "tdiff":
        raise ValueError(f"unknown flow_backend {flow_backend!r}")
    for c in range(pred.shape[1]):
        ref_m = temporal_diff_motion(enc32[:, c:c + 1])
        ch_m = temporal_diff_motion(pred[:, c:c + 1].to(torch.float32))
        sims.append(flow_similarity(ref_m, ch_m))
    return sims


def select_from_similarities(sims: Sequence[float], current_step: int) -> List[int]:
    """SCHED:364-365, 408-437 threshold logic."""
    if current_step < 2:
        return []
    corr = np.array(sims)
    mean, std = np.mean(corr), np.std(corr)
    if current_step <= 10:
        max_replace = 0 if current_step <= 5 else 1
        chans = np.argsort(corr)[:max_replace].tolist()
    else:
        thr = mean - 0.625 * std
        below = [i for i, s in enumerate(corr) if s < thr]
        if len(below) < 2:
            chans = np.argsort(corr)[:2].tolist()
        elif len(below) > 6:
            sc = sorted([(i, corr[i]) for i in below], key=lambda x: x[1])
            chans = [i for i, _ in sc[:6]]
        else:
            chans = below
    chans.sort()
    return chans


def select_motion_related_channels(pred: torch.Tensor, enc: torch.Tensor, current_step: int,
                                   flow_backend: str = "tdiff") -> List[int]:
    """SCHED:338-437."""
    if current_step < 2:
        return []
    return select_from_similarities(channel_similarities(pred, enc, flow_backend), current_step)


# ------------------------------------------------------------------------------------------------------------
# CFG + DSG (PIPE:611, 664-681)
# ------------------------------------------------------------------------------------------------------------
def cfg_combine(cond: torch.Tensor, uncond: torch.Tensor, g: float) -> torch.Tensor:
    """PIPE:611 (note: cond + g*(cond - uncond))."""
    return cond + g * (cond - uncond)


def dsg(good: torch.Tensor, worse: torch.Tensor, omega: float) -> torch.Tensor:
    """PIPE:669-681."""
    dims = list(range(1, good.dim()))
    dot = torch.sum(good * worse, dim=dims, keepdim=True)
    ng = torch.sqrt(torch.sum(good ** 2, dim=dims, keepdim=True))
    nw = torch.sqrt(torch.sum(worse ** 2, dim=dims, keepdim=True))
    cos = dot / (ng * nw + 1e-8)
    ang = torch.acos(torch.clamp(cos, -1.0, 1.0))
    sin = torch.sin(ang)
    ratio = ng / (nw + 1e-8)
    return good + omega * sin * (good - (ratio * cos) * worse)
