"""Tensor-level wrappers over the C-ABI (include/wf_hip.h) for the scheduler / injection set.

PyTorch is used for device memory and streams only: every function takes CUDA(ROCm) tensors, allocates the output with
torch.empty and launches the HIP kernel on torch's current stream.  There is no CPU path here.
"""
from __future__ import annotations

from typing import Optional, Sequence

import threading

import torch

from . import _ffi
from ._ffi import WF_BF16, WF_F32, call


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return WF_F32
    if t.dtype == torch.bfloat16:
        return WF_BF16
    raise TypeError(f"unsupported dtype {t.dtype} (float32 / bfloat16 only)")


def _dev(t: torch.Tensor) -> torch.Tensor:
    if not t.is_cuda:
        raise RuntimeError("worldforge_amd ops need device tensors (there is no CPU fallback)")
    return t if t.is_contiguous() else t.contiguous()


def stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _promote(*ts: torch.Tensor) -> torch.dtype:
    return torch.bfloat16 if all(t.dtype == torch.bfloat16 for t in ts) else torch.float32


def cfg_combine(cond: torch.Tensor, uncond: torch.Tensor, g: float) -> torch.Tensor:
    """PIPE:611."""
    cond, uncond = _dev(cond), _dev(uncond)
    assert cond.dtype == uncond.dtype and cond.shape == uncond.shape
    out = torch.empty_like(cond)
    call("wf_cfg_combine", cond.data_ptr(), uncond.data_ptr(), out.data_ptr(), _dt(cond), float(g), cond.numel(), stream())
    return out


def x0_from_v(sample: torch.Tensor, v: torch.Tensor, sigma: float) -> torch.Tensor:
    """SCHED:958."""
    sample, v = _dev(sample), _dev(v)
    assert sample.shape == v.shape
    out = torch.empty(sample.shape, dtype=_promote(sample, v), device=sample.device)
    call("wf_x0_from_v", sample.data_ptr(), _dt(sample), v.data_ptr(), _dt(v), out.data_ptr(), float(sigma), sample.numel(),
         stream())
    return out


def unipc_update(x: torch.Tensor, m0: torch.Tensor, m1: Optional[torch.Tensor], c1: float, c2: float, c3: float,
                 rk: float) -> torch.Tensor:
    """SCHED:1083-1098."""
    x, m0 = _dev(x), _dev(m0)
    out = torch.empty_like(x)
    if m1 is not None:
        m1 = _dev(m1)
        call("wf_unipc_update", x.data_ptr(), _dt(x), m0.data_ptr(), _dt(m0), m1.data_ptr(), _dt(m1), out.data_ptr(),
             float(c1), float(c2), float(c3), float(rk), x.numel(), stream())
    else:
        call("wf_unipc_update", x.data_ptr(), _dt(x), m0.data_ptr(), _dt(m0), None, 0, out.data_ptr(), float(c1), float(c2),
             float(c3), 1.0, x.numel(), stream())
    return out


def add_noise(x0: torch.Tensor, noise: torch.Tensor, one_minus_sigma: float, sigma: float) -> torch.Tensor:
    """SCHED:1584."""
    x0, noise = _dev(x0), _dev(noise)
    out = torch.empty(x0.shape, dtype=_promote(x0, noise), device=x0.device)
    call("wf_add_noise", x0.data_ptr(), _dt(x0), noise.data_ptr(), _dt(noise), out.data_ptr(), float(one_minus_sigma),
         float(sigma), x0.numel(), stream())
    return out


def _chan_consts(mean: Sequence[float], std: Sequence[float], dtype: torch.dtype):
    m = torch.tensor(list(mean)).to(dtype)
    istd = 1.0 / torch.tensor(list(std)).to(dtype)
    return _ffi.farr(m.float().tolist()), _ffi.farr(istd.float().tolist())


def latent_denorm(z: torch.Tensor, mean, std) -> torch.Tensor:
    """SCHED:1272-1282 / PIPE:742: (z / (1/std) + mean) in z's dtype, returned as fp32."""
    z = _dev(z)
    B, C = z.shape[0], z.shape[1]
    inner = z.numel() // (B * C)
    m, s = _chan_consts(mean, std, z.dtype)
    out = torch.empty(z.shape, dtype=torch.float32, device=z.device)
    call("wf_latent_affine", z.data_ptr(), _dt(z), out.data_ptr(), WF_F32, m, s, 0, B, C, inner, stream())
    return out


def latent_norm(mu: torch.Tensor, mean, std, const_dtype: torch.dtype = torch.float32) -> torch.Tensor:
    """SCHED:1385 / PIPE:351: (mu - mean) * (1/std) with constants held in `const_dtype`; mu fp32 -> fp32."""
    mu = _dev(mu)
    assert mu.dtype == torch.float32
    B, C = mu.shape[0], mu.shape[1]
    inner = mu.numel() // (B * C)
    m, s = _chan_consts(mean, std, const_dtype)
    out = torch.empty_like(mu)
    call("wf_latent_affine", mu.data_ptr(), WF_F32, out.data_ptr(), WF_F32, m, s, 1, B, C, inner, stream())
    return out


PROFILE_BLEND = None  # bench.py sets this to a list: (start, end, bytes) HIP events around every wf_blend_pixels launch (HBM GB/s)


def blend_pixels(ref: torch.Tensor, mask: torch.Tensor, dec: torch.Tensor) -> torch.Tensor:
    """SCHED:1375-1381.  ref [B,3,F,H,W] fp32, mask [B,1,F,H,W] fp32; dec fp32 -> fp32, or dec bf16 (a bf16 VAE module, the LongCat
    entry's: the statements then run in bf16, LongCat SCHED:1152-1164) -> bf16."""
    ref, mask, dec = _dev(ref), _dev(mask), _dev(dec)
    assert ref.dtype == mask.dtype == torch.float32 and dec.dtype in (torch.float32, torch.bfloat16)
    assert ref.shape == dec.shape and mask.shape[1] == 1 and mask.shape[2:] == dec.shape[2:]
    B, C = dec.shape[0], dec.shape[1]
    inner = dec.numel() // (B * C)
    out = torch.empty_like(dec)
    if dec.dtype == torch.bfloat16:
        call("wf_blend_pixels_bf16", ref.data_ptr(), mask.data_ptr(), dec.data_ptr(), out.data_ptr(), B, C, inner, stream())
        return out
    prof = PROFILE_BLEND
    if prof is not None:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    call("wf_blend_pixels", ref.data_ptr(), mask.data_ptr(), dec.data_ptr(), out.data_ptr(), B, C, inner, stream())
    if prof is not None:
        e1.record()
        # algorithmic bytes: read ref (C ch) + mask (1 ch, broadcast) + dec (C ch), write fused (C ch), fp32
        prof.append((e0, e1, 4 * B * inner * (3 * C + 1)))
    return out


def postprocess_video(x: torch.Tensor) -> torch.Tensor:
    """PIPE:744: [C,F,H,W] in [-1,1] -> [F,H,W,C] in [0,1] (fp32; a bf16 video is processed in bf16 and converted, as diffusers does)."""
    x = _dev(x)
    assert x.dtype in (torch.float32, torch.bfloat16) and x.dim() == 4
    C, F, H, W = x.shape
    out = torch.empty((F, H, W, C), dtype=torch.float32, device=x.device)
    if x.dtype == torch.bfloat16:
        call("wf_postprocess_video_bf16", x.data_ptr(), out.data_ptr(), C, F, H, W, stream())
        return out
    call("wf_postprocess_video", x.data_ptr(), out.data_ptr(), C, F, H, W, stream())
    return out


def cast(x: torch.Tensor, dtype: torch.dtype) -> torch.Tensor:
    x = _dev(x)
    if x.dtype == dtype:
        return x
    out = torch.empty(x.shape, dtype=dtype, device=x.device)
    call("wf_cast", x.data_ptr(), _dt(x), out.data_ptr(), _dt(out), x.numel(), stream())
    return out


def channel_swap_(enc: torch.Tensor, pred: torch.Tensor, channels: Sequence[int]) -> torch.Tensor:
    """SCHED:1410-1412 in place on enc."""
    if len(channels) == 0:
        return enc
    assert enc.is_contiguous() and enc.is_cuda
    pred = _dev(pred)
    B, C = enc.shape[0], enc.shape[1]
    inner = enc.numel() // (B * C)
    call("wf_channel_swap", enc.data_ptr(), _dt(enc), pred.data_ptr(), _dt(pred), _ffi.iarr(channels), len(channels), B, C,
         inner, stream())
    return enc


def resize_bilinear2d(x: torch.Tensor, Ho: int, Wo: int) -> torch.Tensor:
    """SCHED:1316-1324 on [..., Hi, Wi] fp32."""
    x = _dev(x)
    assert x.dtype == torch.float32
    Hi, Wi = x.shape[-2:]
    N = x.numel() // (Hi * Wi)
    out = torch.empty(x.shape[:-2] + (Ho, Wo), dtype=torch.float32, device=x.device)
    call("wf_resize_bilinear2d", x.data_ptr(), out.data_ptr(), N, Hi, Wi, Ho, Wo, stream())
    return out


def resize_nearest2d(x: torch.Tensor, Ho: int, Wo: int) -> torch.Tensor:
    """SCHED:1355-1362 on [..., Hi, Wi] fp32."""
    x = _dev(x)
    assert x.dtype == torch.float32
    Hi, Wi = x.shape[-2:]
    N = x.numel() // (Hi * Wi)
    out = torch.empty(x.shape[:-2] + (Ho, Wo), dtype=torch.float32, device=x.device)
    call("wf_resize_nearest2d", x.data_ptr(), out.data_ptr(), N, Hi, Wi, Ho, Wo, stream())
    return out


_WS = {}


def _workspace(key, nfloats, device):
    """Scratch buffer for the multi-launch entry points (DSG, flow metric, Farneback, split attention), one per calling thread: the
    launches of one call must not interleave with another thread's launches on the same buffer (the simulated-rank tests run several
    ranks as threads of one process; a real job has one process per GPU)."""
    k = (key, device, threading.get_ident())
    t = _WS.get(k)
    if t is None or t.numel() < nfloats:
        t = torch.empty(int(nfloats), dtype=torch.float32, device=device)
        _WS[k] = t
    return t


def dsg(good: torch.Tensor, worse: torch.Tensor, omega: float) -> torch.Tensor:
    """PIPE:669-681, one fused call, no host sync."""
    good, worse = _dev(good), _dev(worse)
    assert good.dtype == worse.dtype and good.shape == worse.shape
    ws = _workspace("dsg", _ffi.lib().wf_dsg_workspace_floats(), good.device)
    out = torch.empty_like(good)
    call("wf_dsg", good.data_ptr(), worse.data_ptr(), out.data_ptr(), _dt(good), float(omega), good.numel(), ws.data_ptr(),
         stream())
    return out


def cfg_zero(cond: torch.Tensor, uncond: torch.Tensor, guidance: float, negate: bool = False) -> torch.Tensor:
    """LongCat CFG-zero (pipeline_longcat_video.py:374-383, 875-888) for one sample, one fused call, no host sync."""
    cond, uncond = _dev(cond).contiguous(), _dev(uncond).contiguous()
    assert cond.dtype == uncond.dtype == torch.float32 and cond.shape == uncond.shape
    ws = _workspace("dsg", _ffi.lib().wf_dsg_workspace_floats(), cond.device)
    out = torch.empty_like(cond)
    call("wf_cfg_zero", cond.data_ptr(), uncond.data_ptr(), out.data_ptr(), float(guidance), 1 if negate else 0, cond.numel(),
         ws.data_ptr(), stream())
    return out


def temporal_diff(x: torch.Tensor) -> torch.Tensor:
    """SCHED:391-392: x [C,T,h,w] -> [C,T-1,h,w] fp32."""
    x = _dev(x)
    C, T, h, w = x.shape
    out = torch.empty((C, T - 1, h, w), dtype=torch.float32, device=x.device)
    call("wf_temporal_diff", x.data_ptr(), _dt(x), out.data_ptr(), C, T, h * w, stream())
    return out


_DECAY = {"linear": 0, "exponential": 1, "sine": 2, "cosine": 3}


def soften_mask(mask: torch.Tensor, transition_distance: int = 15, decay_type: str = "sine") -> torch.Tensor:
    """INFER:105-150 on the device: mask [F,H,W] fp32 -> softened [F,H,W] fp32."""
    if decay_type not in _DECAY:
        raise ValueError(f"Unsupported decay type: {decay_type}")
    mask = _dev(mask)
    assert mask.dtype == torch.float32 and mask.dim() == 3 and mask.is_contiguous()
    out = torch.empty_like(mask)
    F_, H, W = mask.shape
    call("wf_soften_mask", mask.data_ptr(), out.data_ptr(), F_, H, W, int(transition_distance), _DECAY[decay_type], stream())
    return out


def farneback_flows(x: torch.Tensor, quant_mode: int = 0) -> torch.Tensor:
    """SCHED:156-248 for every channel at once: x [C,T,h,w] (f32 / bf16) -> flows [C,T-1,2,h,w] fp32 (device).
    quant_mode 0 = the Wan scheduler's uint8 preparation (one global range), 1 = the LongCat scheduler's (a range per channel)."""
    x = _dev(x)
    C, T, h, w = x.shape
    nbytes = _ffi.lib().wf_farneback_workspace_bytes(C, T, h, w)
    ws = _workspace("farneback", (nbytes + 3) // 4, x.device)
    out = torch.empty((C, T - 1, 2, h, w), dtype=torch.float32, device=x.device)
    call("wf_farneback_flows", x.data_ptr(), _dt(x), out.data_ptr(), C, T, h, w, quant_mode, ws.data_ptr(), stream())
    return out


def flow_metrics(ref_flow: torch.Tensor, chan_flow: torch.Tensor, variant: int = 0) -> torch.Tensor:
    """SCHED:497-607 (variant 0) or the LongCat scheduler's metric (variant 1) for n channels at once.
    ref_flow [n,Tm,Cr,h,w], chan_flow [n,Tm,Cc,h,w] fp32 -> sim [n] (device)."""
    ref_flow, chan_flow = _dev(ref_flow), _dev(chan_flow)
    assert ref_flow.dtype == chan_flow.dtype == torch.float32
    n, Tm, Cr, h, w = ref_flow.shape
    Cc = chan_flow.shape[2]
    assert chan_flow.shape == (n, Tm, Cc, h, w)
    ws = _workspace("flow", _ffi.lib().wf_flow_metrics_workspace_floats(n), ref_flow.device)
    sim = torch.empty(n, dtype=torch.float32, device=ref_flow.device)
    call("wf_flow_metrics_variant", ref_flow.data_ptr(), chan_flow.data_ptr(), sim.data_ptr(), n, Tm, Cr, Cc, h * w, variant,
         ws.data_ptr(), stream())
    return sim
