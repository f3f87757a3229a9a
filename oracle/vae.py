"""Oracle: Wan2.1 3D causal VAE encode / decode in plain torch (CPU fp32).  TEST INFRASTRUCTURE ONLY.

Restates /root/reference/wan_for_worldforge/wan/modules/vae.py (the in-tree statement of diffusers' AutoencoderKLWan;
same graph as longcat_video/modules/autoencoder_kl_wan.py:1145-1228) in WHOLE-SEQUENCE form: the reference walks the
video in chunks (1 + 4k frames for encode, one latent frame for decode) and carries the last two frames of every causal
convolution's input in `feat_cache` (vae.py:202-217, 321-331, 516-568).  That is arithmetically a causal (front
zero-padded) temporal convolution over the whole sequence, with three special cases that this file spells out:

  * Resample('upsample3d') (vae.py:103-137): the first latent frame by-passes `time_conv` ('Rep' sentinel); frames 1.. go
    through it as a sequence of their own with causal ZERO padding (the cache that follows 'Rep' is [0, x1]); each of those
    frames yields two (channel halves interleaved in time) -> 1 + 2(T-1) frames.
  * Resample('downsample3d') (vae.py:143-159): frame 0 by-passes `time_conv`; output j >= 1 is the stride-2, 3-tap
    convolution over input frames (2j-2, 2j-1, 2j) -> 1 + (T-1)/2 frames.
  * decode clamps to [-1, 1] (autoencoder_kl_wan.py:1222 / vae.py:661).

Pinned against the imported twin (chunked, cached) in tests/test_oracle_vae.py with goldens tests/golden/g8_vae.npz.
Weights are a flat dict keyed like the twin's state_dict.
"""
from __future__ import annotations

import math
from typing import Dict, List, Tuple

import torch
import torch.nn.functional as F

DIM, Z_DIM = 96, 16
DIM_MULT = [1, 2, 4, 4]
NUM_RES = 2
T_DOWN = [False, True, True]  # vae.py:603
MEAN = [-0.7571, -0.7089, -0.9113, 0.1075, -0.1745, 0.9653, -0.1517, 1.5508, 0.4134, -0.0715, 0.5517, -0.3632, -0.1922,
        -0.9497, 0.2503, -0.2921]  # vae.py:629-632
STD = [2.8184, 1.4541, 2.3275, 2.6558, 1.2196, 1.7708, 2.6052, 2.0743, 3.2687, 2.1526, 2.8652, 1.5579, 1.6382, 1.1253, 2.8251,
       1.9160]  # vae.py:633-636


# ------------------------------------------------------------------------------------------------------------
# architecture plans (vae.py:265-316 encoder, :369-421 decoder)
# ------------------------------------------------------------------------------------------------------------
def encoder_plan(dim=DIM, z=Z_DIM * 2) -> List[Tuple]:
    dims = [dim * u for u in [1] + DIM_MULT]
    plan = [("conv", "encoder.conv1", 3, dims[0])]
    idx = 0
    for i, (cin, cout) in enumerate(zip(dims[:-1], dims[1:])):
        for _ in range(NUM_RES):
            plan.append(("res", f"encoder.downsamples.{idx}", cin, cout))
            idx += 1
            cin = cout
        if i != len(DIM_MULT) - 1:
            plan.append(("down3d" if T_DOWN[i] else "down2d", f"encoder.downsamples.{idx}", cout, cout))
            idx += 1
    c = dims[-1]
    plan += [("res", "encoder.middle.0", c, c), ("attn", "encoder.middle.1", c, c), ("res", "encoder.middle.2", c, c),
             ("head", "encoder.head", c, z)]
    return plan


def decoder_plan(dim=DIM, z=Z_DIM) -> List[Tuple]:
    dims = [dim * u for u in [DIM_MULT[-1]] + DIM_MULT[::-1]]
    t_up = T_DOWN[::-1]
    plan = [("conv", "decoder.conv1", z, dims[0]), ("res", "decoder.middle.0", dims[0], dims[0]),
            ("attn", "decoder.middle.1", dims[0], dims[0]), ("res", "decoder.middle.2", dims[0], dims[0])]
    idx = 0
    for i, (cin, cout) in enumerate(zip(dims[:-1], dims[1:])):
        if i in (1, 2, 3):
            cin = cin // 2
        for _ in range(NUM_RES + 1):
            plan.append(("res", f"decoder.upsamples.{idx}", cin, cout))
            idx += 1
            cin = cout
        if i != len(DIM_MULT) - 1:
            plan.append(("up3d" if t_up[i] else "up2d", f"decoder.upsamples.{idx}", cout, cout // 2))
            idx += 1
    plan.append(("head", "decoder.head", dims[-1], 3))
    return plan


def param_shapes() -> Dict[str, Tuple[int, ...]]:
    """Names / shapes of every parameter, as in WanVAE_(dim=96, z_dim=16).state_dict()."""
    sh: Dict[str, Tuple[int, ...]] = {}

    def conv3(p, cin, cout, k=(3, 3, 3)):
        sh[p + ".weight"] = (cout, cin) + k
        sh[p + ".bias"] = (cout,)

    def conv2(p, cin, cout, k=3):
        sh[p + ".weight"] = (cout, cin, k, k)
        sh[p + ".bias"] = (cout,)

    for plan in (encoder_plan(), decoder_plan()):
        for kind, p, cin, cout in plan:
            if kind == "conv":
                conv3(p, cin, cout)
            elif kind == "res":
                sh[p + ".residual.0.gamma"] = (cin, 1, 1, 1)
                conv3(p + ".residual.2", cin, cout)
                sh[p + ".residual.3.gamma"] = (cout, 1, 1, 1)
                conv3(p + ".residual.6", cout, cout)
                if cin != cout:
                    conv3(p + ".shortcut", cin, cout, (1, 1, 1))
            elif kind == "attn":
                sh[p + ".norm.gamma"] = (cin, 1, 1)
                conv2(p + ".to_qkv", cin, 3 * cin, 1)
                conv2(p + ".proj", cin, cin, 1)
            elif kind in ("down2d", "down3d"):
                conv2(p + ".resample.1", cin, cin)
                if kind == "down3d":
                    conv3(p + ".time_conv", cin, cin, (3, 1, 1))
            elif kind in ("up2d", "up3d"):
                conv2(p + ".resample.1", cin, cin // 2)
                if kind == "up3d":
                    conv3(p + ".time_conv", cin, cin * 2, (3, 1, 1))
            elif kind == "head":
                sh[p + ".0.gamma"] = (cin, 1, 1, 1)
                conv3(p + ".2", cin, cout)
    conv3("conv1", Z_DIM * 2, Z_DIM * 2, (1, 1, 1))
    conv3("conv2", Z_DIM, Z_DIM, (1, 1, 1))
    return sh


def random_weights(seed: int = 0) -> Dict[str, torch.Tensor]:
    """Synthetic weights (no checkpoints offline): fan-in scaled normal convolutions, gamma ~ 1, small biases."""
    g = torch.Generator().manual_seed(seed)
    W = {}
    for k, s in param_shapes().items():
        if k.endswith("gamma"):
            W[k] = 1 + 0.05 * torch.randn(s, generator=g)
        elif k.endswith("bias"):
            W[k] = 0.02 * torch.randn(s, generator=g)
        else:
            fan_in = math.prod(s[1:])
            W[k] = torch.randn(s, generator=g) / math.sqrt(fan_in)
    return W


# ------------------------------------------------------------------------------------------------------------
# whole-sequence building blocks; tensors are [1, C, T, H, W]
# ------------------------------------------------------------------------------------------------------------
def causal_conv3d(x, W, p, stride=(1, 1, 1)):
    """vae.py:17-36 with cache_x=None on the whole sequence: front-pad time by 2*pad_t, pad space symmetrically."""
    w, b = W[p + ".weight"], W[p + ".bias"]
    kt, kh, kw = w.shape[2:]
    x = F.pad(x, (kw // 2, kw // 2, kh // 2, kh // 2, kt - 1, 0))
    return F.conv3d(x, w, b, stride=stride)


def rms_norm(x, gamma):
    """vae.py:39-54 (channel_first): F.normalize(x, dim=1) * sqrt(C) * gamma."""
    return F.normalize(x, dim=1) * (x.shape[1] ** 0.5) * gamma.reshape(1, -1, *([1] * (x.dim() - 2)))


def res_block(x, W, p, cin, cout):
    """vae.py:186-220."""
    h = causal_conv3d(x, W, p + ".shortcut") if cin != cout else x
    y = causal_conv3d(F.silu(rms_norm(x, W[p + ".residual.0.gamma"])), W, p + ".residual.2")
    y = causal_conv3d(F.silu(rms_norm(y, W[p + ".residual.3.gamma"])), W, p + ".residual.6")
    return y + h


def attn_block(x, W, p):
    """vae.py:223-262: per-frame single-head attention over h*w positions."""
    b, c, t, h, w = x.shape
    y = x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w)
    y = rms_norm(y, W[p + ".norm.gamma"])
    qkv = F.conv2d(y, W[p + ".to_qkv.weight"], W[p + ".to_qkv.bias"]).reshape(b * t, 1, 3 * c, h * w).permute(0, 1, 3, 2)
    q, k, v = qkv.chunk(3, dim=-1)
    o = F.scaled_dot_product_attention(q, k, v).squeeze(1).permute(0, 2, 1).reshape(b * t, c, h, w)
    o = F.conv2d(o, W[p + ".proj.weight"], W[p + ".proj.bias"])
    return o.reshape(b, t, c, h, w).permute(0, 2, 1, 3, 4) + x


def _per_frame(x, fn):
    b, c, t, h, w = x.shape
    y = fn(x.permute(0, 2, 1, 3, 4).reshape(b * t, c, h, w))
    return y.reshape(b, t, *y.shape[1:]).permute(0, 2, 1, 3, 4)


def upsample(x, W, p, temporal: bool):
    """vae.py:101-141 ('upsample2d' / 'upsample3d')."""
    if temporal and x.shape[2] > 1:
        c = x.shape[1]
        tail = causal_conv3d(x[:, :, 1:], W, p + ".time_conv")  # frames 1.. only, zero history (the 'Rep' path)
        b, _, t, h, w = tail.shape
        tail = tail.reshape(b, 2, c, t, h, w)
        tail = torch.stack((tail[:, 0], tail[:, 1]), 3).reshape(b, c, 2 * t, h, w)
        x = torch.cat([x[:, :, :1], tail], dim=2)
    return _per_frame(x, lambda y: F.conv2d(F.interpolate(y, scale_factor=(2.0, 2.0), mode="nearest-exact"),
                                            W[p + ".resample.1.weight"], W[p + ".resample.1.bias"], padding=1))


def downsample(x, W, p, temporal: bool):
    """vae.py:87-96, 139-159 ('downsample2d' / 'downsample3d')."""
    x = _per_frame(x, lambda y: F.conv2d(F.pad(y, (0, 1, 0, 1)), W[p + ".resample.1.weight"], W[p + ".resample.1.bias"],
                                         stride=2))
    if temporal and x.shape[2] > 1:
        tail = F.conv3d(x, W[p + ".time_conv.weight"], W[p + ".time_conv.bias"], stride=(2, 1, 1))  # windows (0,1,2),(2,3,4),..
        x = torch.cat([x[:, :, :1], tail], dim=2)
    return x


def run_plan(x, W, plan):
    for kind, p, cin, cout in plan:
        if kind == "conv":
            x = causal_conv3d(x, W, p)
        elif kind == "res":
            x = res_block(x, W, p, cin, cout)
        elif kind == "attn":
            x = attn_block(x, W, p)
        elif kind in ("down2d", "down3d"):
            x = downsample(x, W, p, kind == "down3d")
        elif kind in ("up2d", "up3d"):
            x = upsample(x, W, p, kind == "up3d")
        elif kind == "head":
            x = causal_conv3d(F.silu(rms_norm(x, W[p + ".0.gamma"])), W, p + ".2")
    return x


def encode_mode(W, video: torch.Tensor) -> torch.Tensor:
    """vae.encode(x).latent_dist.mode() (SCHED:1384, PIPE:348): [1,3,F,H,W] -> posterior mean [1,16,T,h,w] (un-normalised)."""
    assert (video.shape[2] - 1) % 4 == 0
    out = run_plan(video.float(), W, encoder_plan())
    mu, _ = causal_conv3d(out, W, "conv1").chunk(2, dim=1)  # vae.py:535
    return mu


def encode_sample(W, video: torch.Tensor, generator=None) -> torch.Tensor:
    """vae.encode(x).latent_dist.sample(generator) (LongCat prepare_latents, pipeline_longcat_video.py:278; diffusers'
    DiagonalGaussianDistribution: logvar clamped to [-30, 20], mean + exp(logvar / 2) * randn)."""
    assert (video.shape[2] - 1) % 4 == 0
    out = run_plan(video.float(), W, encoder_plan())
    mu, logvar = causal_conv3d(out, W, "conv1").chunk(2, dim=1)
    std = torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0))
    return mu + std * torch.randn(mu.shape, generator=generator, dtype=mu.dtype)


def decode(W, z: torch.Tensor) -> torch.Tensor:
    """vae.decode(z)[0] (SCHED:1285, PIPE:743): [1,16,T,h,w] (de-normalised) -> [1,3,4T-3,8h,8w], clamped to [-1,1]."""
    x = causal_conv3d(z.float(), W, "conv2")  # vae.py:553
    return run_plan(x, W, decoder_plan()).clamp(-1.0, 1.0)
