"""Wan2.1 3D causal VAE (AutoencoderKLWan) on hand-written HIP kernels.

Speaks the diffusers protocol the reference uses (SCHED:1272-1285, 1384; PIPE:162-163, 348, 743):
    vae.encode(x[B,3,F,H,W]).latent_dist.mode() -> [B,16,T,h,w]      vae.decode(z[B,16,T,h,w], return_dict=False)[0]
    vae.config.{z_dim, latents_mean, latents_std}, vae.dtype, vae.temperal_downsample
The arithmetic follows /root/reference/wan_for_worldforge/wan/modules/vae.py (in-tree statement of the diffusers class).

MI355X-first design: instead of the reference's 21 sequential per-latent-frame decoder passes with a Python-side
feature cache, the WHOLE frame sequence stays resident in HBM (channels-last [T,H,W,C]; 81x480x832x96 fp32 = 12.4 GB,
trivial against 288 GB) and every layer is ONE launch over all frames with causal zero padding in time -- arithmetically
the same function (oracle/vae.py proves it against the chunked twin).  Convolutions with MFMA-sized channel counts run as
implicit GEMM (csrc/conv.hip); RMS-norm + SiLU is a fused one-pass kernel producing the bf16 conv operand; the nearest
2x upsample and the stride-2 / zero-pad of Resample are folded into the convolution's gather; the mid-block attention is
two MFMA GEMMs around a row softmax.  fp32 residual stream, fp32 accumulation.

precision (the reference loads the VAE with torch_dtype=torch.float32, INFER:185-189):
  "bf16"  -- every matrix-core operand (activation and weight) rounded to bf16: 2^-9 relative per operand.  Opt-in fast mode: over a
             guided job's 31 decode -> encode round trips this noise is enough to flip near-tied FLF gate decisions
             (tools/vae_precision_study.py).
  "fp16x3" -- DEFAULT since round 4 (alias "fp32").  The three-term split below on FP16 parts (hi = fp16(x), lo = fp16(x - hi): 22
             significand bits) on v_mfma_f32_32x32x16_f16: same kernels, same cost as the bf16 split.  Per product ~2^-22 WHERE lo IS A
             NORMAL fp16, i.e. |x| >= 2^-3; below that lo is a subnormal with an absolute floor of 2^-25.  Round 5: every weight matrix is
             stored times an exact power of two that lifts it into the normal range (undone in the kernels' epilogues, `acc_scale`), so
             the floor only remains for small ACTIVATIONS, whose absolute error 2^-25 |w| is negligible against the products of the O(1)
             ones.  Measured end to end against the fp32 goldens: see tests/test_gpu_vae.py (rel. L2) -- that, not the per-product
             figure, is the claim.  fp16 cannot hold |x| > 65504: the producers raise a device flag; it is copied to the host
             asynchronously and turned into a RuntimeError at the next check point (check_range; never a silent inf) -- such weights
             need "bf16x3".
  "fp16"  -- round 6 (alias "tf32"): ONE fp16 term per operand (hi = fp16(x): 10 explicit mantissa bits, fp32 accumulation) on the same
             *_f16 kernels with the same power-of-two weight scale and range flag -- the multiplicand width of a TF32 convolution, which
             is what an fp32 `conv3d` executes under PyTorch's defaults on the reference's stack (torch.backends.cudnn.allow_tf32 = True;
             the LongCat entry sets it explicitly, run_longcat_worldforge_single.py:144-146).  One third of "fp16x3"'s matrix work.  The
             tensors between layers stay f32 (residual stream, norm inputs, attention scores and probabilities), as in the three-term
             modes.  Opt-in for the Wan path (the headline keeps "fp16x3": BASELINE.md states an fp32 VAE and the parity target is the
             CPU eager path).
  "bf16x3" -- the default of rounds 2-3.  fp32-CLASS contractions on the bf16 matrix cores, NOT IEEE fp32: every operand x is carried as hi = bf16(x),
             lo = bf16(x - hi) and every contraction as hi.hi + lo.hi + hi.lo in fp32 accumulators (wf_split_bf16x3: activations
             [hi | lo | hi], weights [hi | hi | lo] on 3x the channels, the SAME conv / GEMM kernels; the dropped lo.lo term and the
             rounding of lo leave ~2^-16 relative per product, against 2^-24 for true fp32).  3x the MFMA work.  "fp32" is accepted as
             an alias (rounds 1-2 called the mode that) and maps to "bf16x3".
"""
from __future__ import annotations

import os

import math
from types import SimpleNamespace
from typing import Dict, List, Tuple

import torch

from . import ops
from . import _ffi
from ._ffi import WF_BF16, WF_F32, call
from .dit import EPI_BF16, EPI_F32, EPI_F32_ACC, gemm

DIM, Z_DIM = 96, 16
DIM_MULT = [1, 2, 4, 4]
NUM_RES = 2
T_DOWN = [False, True, True]
LATENTS_MEAN = [-0.7571, -0.7089, -0.9113, 0.1075, -0.1745, 0.9653, -0.1517, 1.5508, 0.4134, -0.0715, 0.5517, -0.3632,
                -0.1922, -0.9497, 0.2503, -0.2921]  # vae.py:629-632
LATENTS_STD = [2.8184, 1.4541, 2.3275, 2.6558, 1.2196, 1.7708, 2.6052, 2.0743, 3.2687, 2.1526, 2.8652, 1.5579, 1.6382,
               1.1253, 2.8251, 1.9160]  # vae.py:633-636

BF, F32 = torch.bfloat16, torch.float32


def encoder_plan() -> List[Tuple]:
    """vae.py:283-316."""
    dims = [DIM * u for u in [1] + DIM_MULT]
    plan = [("conv_in", "encoder.conv1", 3, dims[0])]
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
             ("head", "encoder.head", c, 2 * Z_DIM)]
    return plan


def decoder_plan() -> List[Tuple]:
    """vae.py:387-421."""
    dims = [DIM * u for u in [DIM_MULT[-1]] + DIM_MULT[::-1]]
    t_up = T_DOWN[::-1]
    plan = [("conv_in", "decoder.conv1", Z_DIM, dims[0]), ("res", "decoder.middle.0", dims[0], dims[0]),
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


def diffusers_key_map() -> Dict[str, str]:
    """diffusers `AutoencoderKLWan` module name -> in-tree twin (`WanVAE_`) module name, for the Wan 2.1 configuration
    (is_residual=False).  The diffusers class is what the reference executes (INFER:185-189 `AutoencoderKLWan.from_pretrained`; a
    vendored copy is longcat_video/modules/autoencoder_kl_wan.py: encoder :505-584, decoder :783-872, mid block :430-451, up block
    :714-753, residual block :311-340, top level :1029-1052).  Leaves (.weight / .bias / .gamma) are appended by the caller."""
    m = {"encoder.conv_in": "encoder.conv1", "encoder.norm_out": "encoder.head.0", "encoder.conv_out": "encoder.head.2",
         "decoder.conv_in": "decoder.conv1", "decoder.norm_out": "decoder.head.0", "decoder.conv_out": "decoder.head.2",
         "quant_conv": "conv1", "post_quant_conv": "conv2"}
    res = {"norm1": "residual.0", "conv1": "residual.2", "norm2": "residual.3", "conv2": "residual.6", "conv_shortcut": "shortcut"}
    for side in ("encoder", "decoder"):
        for j, name in ((0, "resnets.0"), (1, "attentions.0"), (2, "resnets.1")):
            d, t = f"{side}.mid_block.{name}", f"{side}.middle.{j}"
            if j == 1:
                for leaf in ("norm", "to_qkv", "proj"):
                    m[f"{d}.{leaf}"] = f"{t}.{leaf}"
            else:
                for a, b in res.items():
                    m[f"{d}.{a}"] = f"{t}.{b}"
    # encoder: down_blocks is the same flat list as the twin's downsamples
    for kind, p, cin, cout in encoder_plan():
        if not p.startswith("encoder.downsamples."):
            continue
        d = p.replace("encoder.downsamples.", "encoder.down_blocks.")
        if kind == "res":
            for a, b in res.items():
                if a != "conv_shortcut" or cin != cout:
                    m[f"{d}.{a}"] = f"{p}.{b}"
        else:
            m[f"{d}.resample.1"] = f"{p}.resample.1"
            if kind.endswith("3d"):
                m[f"{d}.time_conv"] = f"{p}.time_conv"
    # decoder: up_blocks[i] = {resnets[0..2], upsamplers[0]} against the twin's flat upsamples list
    blk, j = 0, 0
    for kind, p, cin, cout in decoder_plan():
        if not p.startswith("decoder.upsamples."):
            continue
        if kind == "res":
            d = f"decoder.up_blocks.{blk}.resnets.{j}"
            for a, b in res.items():
                if a != "conv_shortcut" or cin != cout:
                    m[f"{d}.{a}"] = f"{p}.{b}"
            j += 1
            if j == NUM_RES + 1 and blk == len(DIM_MULT) - 1:
                blk, j = blk + 1, 0
        else:
            d = f"decoder.up_blocks.{blk}.upsamplers.0"
            m[f"{d}.resample.1"] = f"{p}.resample.1"
            if kind.endswith("3d"):
                m[f"{d}.time_conv"] = f"{p}.time_conv"
            blk, j = blk + 1, 0
    return m


def diffusers_to_twin_state_dict(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """Rename a diffusers-layout AutoencoderKLWan state dict to the twin's names (tensors are shared, shapes are the same up to the
    trailing singleton dims of the RMS-norm gammas, which the loader flattens).  Unknown or missing parameters raise KeyError."""
    km = diffusers_key_map()
    out = {}
    for k, v in sd.items():
        base, _, leaf = k.rpartition(".")
        if base not in km:
            raise KeyError(f"unmapped diffusers AutoencoderKLWan parameter {k!r}")
        out[f"{km[base]}.{leaf}"] = v
    return out


class _LatentDist:
    """diffusers' DiagonalGaussianDistribution surface the samplers use: mode() (Wan path, PIPE retrieve_latents "argmax") and
    sample(generator) (LongCat prepare_latents, pipeline_longcat_video.py:278: mean + std * randn, logvar clamped to [-30, 20])."""

    def __init__(self, mean, logvar=None):
        self._mean = mean
        self._logvar = logvar

    def mode(self):
        return self._mean

    def sample(self, generator=None):
        std = torch.exp(0.5 * torch.clamp(self._logvar, -30.0, 20.0))
        if generator is not None and generator.device.type != self._mean.device.type:
            noise = torch.randn(self._mean.shape, generator=generator, dtype=self._mean.dtype).to(self._mean.device)
        else:
            noise = torch.randn(self._mean.shape, generator=generator, dtype=self._mean.dtype, device=self._mean.device)
        return self._mean + std * noise


class AutoencoderKLWan:
    dtype = torch.float32

    ATTN_BATCH_BYTES = 12 << 30  # scores, probabilities and split operands of the mid-block attention's batched launches kept at a time (_attn_x3)
    crop_to_mask = True          # the IRR injection decodes only the pixel columns its blend can see (needed_columns); False: everything

    def __init__(self, device="cuda:0", comm=None, precision: str = "fp16x3", dtype: torch.dtype = torch.float32, strict_range: bool = False):
        """strict_range: check the fp16 range flag INSIDE the call that raised it (one host synchronisation per encode / decode) -- for
        callers of the bare encode / decode API who consume the result without ever calling check_range() or the VAE again.
        dtype: the MODULE dtype the protocol shows (`vae.dtype`, what `from_pretrained(torch_dtype=...)` sets): torch.float32 for the
        Wan entry (INFER:185-189), torch.bfloat16 for the LongCat entry (run_longcat_worldforge_single.py:205).  A bf16 module takes bf16
        inputs (others are rounded to bf16 on the way in, the `.to(dtype=vae.dtype)` of LongCat's fuse_latents) and returns bf16 videos /
        moments; in between, this implementation keeps its f32 stream with `precision` operands -- closer to the fp32 network than the
        eager bf16 module, whose every activation is bf16 (tests/golden/g8c)."""
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError(f"dtype must be torch.float32 or torch.bfloat16, got {dtype}")
        self.dtype = dtype
        self.strict_range = bool(strict_range)
        if precision == "fp32":  # the fp32-CLASS mode of the round: three-term split operands, fp16 parts since round 4
            precision = "fp16x3"
        if precision == "tf32":  # what the mode stands for: the multiplicand width of a TF32 convolution
            precision = "fp16"
        if precision not in ("bf16", "bf16x3", "fp16x3", "fp16"):
            raise ValueError(f"precision must be 'fp16x3' (alias 'fp32'), 'bf16x3', 'fp16' (alias 'tf32') or 'bf16', got {precision!r}")
        self.precision = precision
        self.x3 = precision in ("bf16x3", "fp16x3")  # three-term split operands
        self.f16 = precision in ("fp16x3", "fp16")     # operand parts are fp16 (v_mfma_f32_32x32x16_f16) instead of bf16
        # `wide`: every tensor BETWEEN two layers is f32 and the 16-bit operand is made by a producer that checks the fp16 range (the
        # three-term modes, and the one-term fp16 mode); only "bf16" takes the 16-bit output copies of the conv / GEMM epilogues
        self.wide = self.x3 or self.f16
        self.terms = 3 if self.x3 else 1
        self.OP = torch.float16 if self.f16 else BF    # element type of every matrix-core operand tensor
        self._sfx = "_f16" if self.f16 else ""         # the C-ABI entry points of that element type
        self.device = torch.device(device)
        self.comm = comm  # row-slab sharding of the high-resolution stages over the ranks of `comm` (parallel.Comm or a stand-in)
        self._reps = 1    # ranks per row group while a sharded stage runs (see _row_groups)
        self.config = SimpleNamespace(z_dim=Z_DIM, latents_mean=LATENTS_MEAN, latents_std=LATENTS_STD)
        self.temperal_downsample = list(T_DOWN)
        self.w: Dict[str, torch.Tensor] = {}
        self.flops_last = 0

    # ------------------------------------------------------------------------------------------------------------
    # weights (keys as in WanVAE_.state_dict())
    # ------------------------------------------------------------------------------------------------------------
    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        dev = self.device
        W: Dict[str, torch.Tensor] = {}
        x3 = self.x3

        # W[key + ".scale"] (a Python float, fp16x3 only; absent = 1): acc_scale = 2^-k of a weight operand stored scaled by 2^k -- kept
        # in the weight dict so that instances sharing `w` share it

        def operand(w, key):  # f32 [..., Cin] (host) -> the matrix-core weight operand on the device
            w = w.to(F32)
            if not self.wide:
                return w.to(device=dev, dtype=BF).contiguous()
            if self.f16:
                # hi = fp16(w), lo = fp16(w - hi): lo is an fp16 SUBNORMAL (absolute floor 2^-25) for |w| < 2^-3 -- VAE weights are ~0.02, so the
                # split carried them to ~2^-18 relative, not the 2^-22 of its normal range.  Store the matrix times an exact
                # power of two that puts its largest magnitude into [2^13, 2^14) (every weight above 2^-17 of it then has a normal lo; fp16 tops
                # out at 2^16) and hand the kernels acc_scale = 2^-k: out = acc * 2^-k + bias, exact.
                mx = float(w.abs().max())
                if not math.isfinite(mx):
                    raise ValueError("a VAE weight is not finite")
                if mx > 0.0:
                    k = 13 - math.floor(math.log2(mx))
                    k = max(-126, min(126, k))
                    w = w * (2.0 ** k)
                    W[key + ".scale"] = 2.0 ** -k
            hi = w.to(self.OP)
            if not x3:   # "fp16": the one-term operand (the scaled weight can no longer leave the fp16 range)
                return hi.to(dev).contiguous()
            lo = (w - hi.to(F32)).to(self.OP)
            return torch.cat([hi, hi, lo], dim=-1).to(dev).contiguous()  # weight side of wf_split_bf16x3 / wf_split_f16x3

        def mfma_conv(p):  # [Cout,Cin,kt,kh,kw] -> bf16 [Cout, taps, Cin]
            w = sd[p + ".weight"]
            if w.dim() == 4:
                w = w.unsqueeze(2)
            co, ci = w.shape[:2]
            W[p + ".w"] = operand(w.permute(0, 2, 3, 4, 1).reshape(co, -1, ci), p + ".w")
            W[p + ".b"] = sd[p + ".bias"].to(device=dev, dtype=F32).contiguous()

        def mfma_conv_padded(p, cin_pad=None, cout_pad=None):
            """Thin layers on the MFMA path: zero-pad Cin to a 32-channel K slice / Cout to a multiple of 4 (the kernel
            computes a 96-wide tile anyway and only stores Cout columns)."""
            w = sd[p + ".weight"].to(F32)
            b = sd[p + ".bias"].to(F32)
            co, ci = w.shape[:2]
            cip, cop = cin_pad or ci, cout_pad or co
            wp = torch.zeros((cop, cip) + tuple(w.shape[2:]), dtype=F32)
            wp[:co, :ci] = w
            bp = torch.zeros(cop, dtype=F32)
            bp[:co] = b
            W[p + ".w"] = operand(wp.permute(0, 2, 3, 4, 1).reshape(cop, -1, cip), p + ".w")
            W[p + ".b"] = bp.to(dev)

        def small_conv(p, cout_pad=None):  # -> f32 [taps, Cin, Cout]
            w = sd[p + ".weight"].to(F32)
            b = sd[p + ".bias"].to(F32)
            co, ci = w.shape[:2]
            cop = cout_pad or co
            wp = torch.zeros((cop, ci) + tuple(w.shape[2:]), dtype=F32)
            wp[:co] = w
            bp = torch.zeros(cop, dtype=F32)
            bp[:co] = b
            W[p + ".w"] = wp.permute(2, 3, 4, 1, 0).reshape(-1, ci, cop).to(dev).contiguous()
            W[p + ".b"] = bp.to(dev)

        def lin(p):  # 1x1(x1) conv as GEMM weight bf16 [Cout, Cin]
            w = sd[p + ".weight"]
            W[p + ".w"] = operand(w.reshape(w.shape[0], w.shape[1]), p + ".w")
            W[p + ".b"] = sd[p + ".bias"].to(device=dev, dtype=F32).contiguous()

        def gamma(p):
            W[p] = sd[p].reshape(-1).to(device=dev, dtype=F32).contiguous()

        def up_phases(p):
            """Nearest-2x upsample + 3 x 3 conv (vae.py:76-86) as four 2 x 2 convolutions on the source grid: output (2y + py, 2x + px) reads
            source rows {y - 1 + py, y + py}; the 3 x 3 taps landing on the same source pixel are summed here, in fp32, before the operand
            split -- 4 / 9 of the multiply-adds, same result up to fp32 re-association."""
            w = sd[p + ".weight"].to(F32)
            if w.dim() == 5:
                w = w[:, :, 0]
            groups = (([0], [1, 2]), ([0, 1], [2]))  # [phase][source offset] -> the 3-tap indices that read it
            for py in range(2):
                for px in range(2):
                    wp = torch.stack([torch.stack([sum(w[:, :, dy, dx] for dy in groups[py][a] for dx in groups[px][b]) for b in range(2)], dim=-1)
                                      for a in range(2)], dim=-2)  # [Cout, Cin, 2, 2]
                    W[f"{p}.ph{py}{px}.w"] = operand(wp.permute(0, 2, 3, 1).reshape(wp.shape[0], 4, wp.shape[1]), f"{p}.ph{py}{px}.w")

        first_up = True
        for plan in (encoder_plan(), decoder_plan()):
            for kind, p, cin, cout in plan:
                if kind == "conv_in":
                    mfma_conv_padded(p, cin_pad=32)
                elif kind == "res":
                    gamma(p + ".residual.0.gamma")
                    mfma_conv(p + ".residual.2")
                    gamma(p + ".residual.3.gamma")
                    mfma_conv(p + ".residual.6")
                    if cin != cout:
                        lin(p + ".shortcut")
                elif kind == "attn":
                    gamma(p + ".norm.gamma")
                    lin(p + ".to_qkv")
                    lin(p + ".proj")
                elif kind in ("down2d", "down3d", "up2d", "up3d"):
                    mfma_conv(p + ".resample.1")
                    if kind.startswith("up"):
                        # every upsampling conv but the first (small, and its row slabs of the sharded decoder may start on an odd row) runs
                        # as four phase convolutions
                        if not first_up:
                            up_phases(p + ".resample.1")
                        first_up = False
                    if kind.endswith("3d"):
                        mfma_conv(p + ".time_conv")
                elif kind == "head":
                    gamma(p + ".0.gamma")
                    mfma_conv_padded(p + ".2", cout_pad=(cout + 31) // 32 * 32)  # one 32-channel output block of the 3x3x3 kernel
        small_conv("conv1")
        small_conv("conv2", cout_pad=32)  # decoder input, zero-padded to one 32-channel MFMA K slice
        self.w = W
        return self

    def load_diffusers_state_dict(self, sd: Dict[str, torch.Tensor]):
        """State dict in the layout of the class the reference executes (diffusers AutoencoderKLWan, INFER:185-189)."""
        return self.load_state_dict(diffusers_to_twin_state_dict(sd))

    @classmethod
    def from_pretrained(cls, path: str, device="cuda:0", comm=None, precision: str = "fp16x3", subfolder: str = "vae",
                        torch_dtype: torch.dtype = torch.float32):
        """`AutoencoderKLWan.from_pretrained(model_id, subfolder="vae", torch_dtype=torch.float32)` (INFER:185-189) from a local
        diffusers checkpoint directory: reads `<path>/<subfolder>/*.safetensors` (sharded or not) with checkpoint.load_dir."""
        from . import checkpoint
        folder = os.path.join(path, subfolder) if subfolder and os.path.isdir(os.path.join(path, subfolder)) else path
        return cls(device, comm=comm, precision=precision, dtype=torch_dtype).load_diffusers_state_dict(checkpoint.load_dir(folder))

    def init_random(self, seed: int = 0):
        """Synthetic weights of the real shapes, generated on the host in twin layout (127 M parameters)."""
        g = torch.Generator().manual_seed(seed)
        sd = {}

        def conv(p, cin, cout, k):
            sd[p + ".weight"] = torch.randn((cout, cin) + k, generator=g) / math.sqrt(cin * math.prod(k))
            sd[p + ".bias"] = 0.02 * torch.randn(cout, generator=g)

        for plan in (encoder_plan(), decoder_plan()):
            for kind, p, cin, cout in plan:
                if kind == "conv_in":
                    conv(p, cin, cout, (3, 3, 3))
                elif kind == "res":
                    sd[p + ".residual.0.gamma"] = 1 + 0.05 * torch.randn(cin, generator=g)
                    conv(p + ".residual.2", cin, cout, (3, 3, 3))
                    sd[p + ".residual.3.gamma"] = 1 + 0.05 * torch.randn(cout, generator=g)
                    conv(p + ".residual.6", cout, cout, (3, 3, 3))
                    if cin != cout:
                        conv(p + ".shortcut", cin, cout, (1, 1, 1))
                elif kind == "attn":
                    sd[p + ".norm.gamma"] = 1 + 0.05 * torch.randn(cin, generator=g)
                    conv(p + ".to_qkv", cin, 3 * cin, (1, 1))
                    conv(p + ".proj", cin, cin, (1, 1))
                elif kind in ("down2d", "down3d"):
                    conv(p + ".resample.1", cin, cin, (3, 3))
                    if kind == "down3d":
                        conv(p + ".time_conv", cin, cin, (3, 1, 1))
                elif kind in ("up2d", "up3d"):
                    conv(p + ".resample.1", cin, cin // 2, (3, 3))
                    if kind == "up3d":
                        conv(p + ".time_conv", cin, cin * 2, (3, 1, 1))
                elif kind == "head":
                    sd[p + ".0.gamma"] = 1 + 0.05 * torch.randn(cin, generator=g)
                    conv(p + ".2", cin, cout, (3, 3, 3))
        conv("conv1", 2 * Z_DIM, 2 * Z_DIM, (1, 1, 1))
        conv("conv2", Z_DIM, Z_DIM, (1, 1, 1))
        return self.load_state_dict(sd)

    # ------------------------------------------------------------------------------------------------------------
    # kernels wrappers; activations are channels-last [T, H, W, C]
    # ------------------------------------------------------------------------------------------------------------
    def _conv(self, x, p, To, Ho, Wo, Cout, k, st=1, ss=1, pt=0, ps=0, up2=False, tsplit=False, resid=None, out_f32=True,
              out_bf16=False, out_shape=None, out_bf_tensor=None, ph=None):
        """ps: symmetric spatial padding (top = left); ph overrides the top padding (row slabs with halo rows).  x: channels-last
        [T,H,W,C], or the slice-major operand [T,H,C_stored/16,W,16] of _rms(blocked=True) (3x3x3 stride-1 layers only)."""
        layout, Cst = 0, x.shape[-1]
        if x.dim() == 5:
            Ti, Hi, nsl, Wi, _ = x.shape
            layout, Cst = 1, 16 * nsl
            Cin = Cst * 3 // 2 if self.x3 else Cst  # fp32-class: [hi | lo] stored, K = [hi | lo | hi]
        else:
            Ti, Hi, Wi, Cin = x.shape
        assert x.dtype == self.OP and x.is_contiguous()
        shape = out_shape or (To, Ho, Wo, Cout)
        of = torch.empty(shape, dtype=F32, device=x.device) if out_f32 else None
        ob = out_bf_tensor if out_bf_tensor is not None else (torch.empty(shape, dtype=self.OP, device=x.device) if out_bf16 else None)
        W = self.w
        if (tuple(k) == (3, 3, 3) and st == 1 and ss == 1 and pt == 2 and ps == 1 and not up2 and not tsplit and To == Ti and Wo == Wi
                and Cin % 32 == 0 and Cout % 32 == 0):
            # the FLOP-heavy layers: LDS-resident input patch kernel on re-packed weights (packed once per layer, cached).  Chosen by
            # layer type only, never by size: a row slab of the sharded VAE must run the same arithmetic as the whole image
            zp = self._zero_page(int(_ffi.lib().wf_conv3d_333_zero_page_bytes(Wi, Cst, layout)))
            call("wf_conv3d_333" + self._sfx, x.data_ptr(), self._packed333(p, Cout, Cin).data_ptr(), W[p + ".b"].data_ptr(),
                 resid.data_ptr() if resid is not None else None, of.data_ptr() if of is not None else None,
                 ob.data_ptr() if ob is not None else None, Ti, Hi, Wi, Cin, Ho, Cout, ps if ph is None else ph,
                 zp.data_ptr(), zp.numel() * 2, layout, Cst, *self._acc_scale(p + ".w"), ops.stream())
        else:
            assert layout == 0, "slice-major operands are for the 3x3x3 stride-1 kernel only"
            call("wf_conv3d_cl" + self._sfx, x.data_ptr(), W[p + ".w"].data_ptr(), W[p + ".b"].data_ptr(),
                 resid.data_ptr() if resid is not None else None, of.data_ptr() if of is not None else None,
                 ob.data_ptr() if ob is not None else None, Ti, Hi, Wi, Cin, To, Ho, Wo, Cout, k[0], k[1], k[2], st, ss, pt,
                 ps if ph is None else ph, ps, 1 if up2 else 0, 1 if tsplit else 0, self._zero_page().data_ptr(), *self._acc_scale(p + ".w"),
                 ops.stream())
        self.flops_last += 2 * To * Ho * Wo * Cout * k[0] * k[1] * k[2] * Cin
        return of, ob

    def _acc_scale(self, wkey: str):
        """The extra trailing argument of the *_f16 entry points that take a weight operand: () for the bf16 ones."""
        return (float(self.w.get(wkey + ".scale", 1.0)),) if self.f16 else ()

    def _packed333(self, p, Cout, Cin):
        """[Cout][27][Cin] -> [27][Cin/16][Cout][16] copy of a 3x3x3 weight for wf_conv3d_333 (keyed by the weight tensor's identity so
        that instances sharing `w` and reloaded weights stay consistent)."""
        w = self.w[p + ".w"]
        cache = self.__dict__.setdefault("_packed", {})
        hit = cache.get(p)
        if hit is None or hit[0] is not w:
            out = torch.empty((27, Cin // 16, Cout, 16), dtype=w.dtype, device=w.device)
            call("wf_conv3d_pack333", w.data_ptr(), out.data_ptr(), Cout, Cin, ops.stream())
            cache[p] = hit = (w, out)
        return hit[1]

    def _zero_page(self, nbytes: int = 128):
        """Zeros for the convolutions' padding taps.  wf_conv3d_333 reads it at the wave-uniform slice offset of its LDS-DMA pieces
        (wf_conv3d_333_zero_page_bytes: up to (Cin_stored / 16 - 1) x W x 32 bytes), so the page grows to the largest request seen."""
        z = getattr(self, "_zeros", None)
        if z is None or z.numel() * 2 < nbytes:
            z = self._zeros = torch.zeros(max(1 << 20, (nbytes + 1) // 2), dtype=BF, device=self.device)
        return z

    def _small_conv(self, x, p, To, Ho, Wo, Cout, k, pt=0, ps=0, clamp=0.0, out_dtype=F32):
        Ti, Hi, Wi, Cin = x.shape
        out = torch.empty((To, Ho, Wo, Cout), dtype=out_dtype, device=x.device)
        W = self.w
        call("wf_conv3d_small", x.data_ptr(), WF_BF16 if x.dtype == BF else WF_F32, W[p + ".w"].data_ptr(), W[p + ".b"].data_ptr(),
             out.data_ptr() if out_dtype == F32 else None, out.data_ptr() if out_dtype == BF else None, Ti, Hi, Wi, Cin, To, Ho,
             Wo, Cout, k[0], k[1], k[2], 1, 1, pt, ps, float(clamp), ops.stream())
        self.flops_last += 2 * To * Ho * Wo * Cout * k[0] * k[1] * k[2] * Cin
        return out

    def _rms(self, x, gamma, silu=True, blocked=False, halo=False):
        """RMS_norm (+ SiLU) of the f32 stream -> the next layer's matrix-core operand.  blocked: the slice-major operand of the 3x3x3
        kernel, [T,H,C/16,W,16] (fp32-class mode: [hi | lo] slices, [T,H,2C/16,W,16])."""
        C = x.shape[-1]
        if blocked and C % 32 == 0:
            T, H, Wd, _ = x.shape
            # halo: the kernel writes rows 1 .. H of a [T, H + 2, ...] slab operand directly (row slabs: _halo_fill adds the neighbours' rows)
            out = torch.empty((T, H + (2 if halo else 0), (2 if self.x3 else 1) * C // 16, Wd, 16), dtype=self.OP, device=x.device)
            call("wf_rms_silu_cl_blocked" + self._sfx, x.data_ptr(), gamma.data_ptr(), out.data_ptr(), x.numel() // C, C, 1 if silu else 0, Wd,
                 1 if self.x3 else 0, H if halo else 0, ops.stream())
            return out
        if self.x3:
            out = torch.empty(tuple(x.shape[:-1]) + (3 * C,), dtype=self.OP, device=x.device)
            call("wf_rms_silu_cl_x3" + self._sfx, x.data_ptr(), gamma.data_ptr(), out.data_ptr(), x.numel() // C, C, 1 if silu else 0, ops.stream())
            return out
        out = torch.empty(x.shape, dtype=self.OP, device=x.device)
        call("wf_rms_silu_cl" + self._sfx, x.data_ptr(), gamma.data_ptr(), out.data_ptr(), None, x.numel() // C, C, 1 if silu else 0,
             ops.stream())
        return out

    def _operand(self, x, side=0, out=None):
        """f32 [..., C] -> the matrix-core operand: bf16 [..., C], or the three-term split [..., 3C] (precision="fp32"; side 0 =
        activation, 1 = weight-side layout).  `out`: a contiguous destination of the operand's shape."""
        if not self.wide:
            y = ops.cast(x, BF)
            if out is not None:
                out.copy_(y)
                return out
            return y
        assert x.dtype == F32 and x.stride(-1) == 1
        C = x.shape[-1]
        if x.dim() > 2:
            assert x.is_contiguous()
            x2 = x.view(-1, C)
        else:
            x2 = x
        n = self.terms
        if out is None:
            out = torch.empty(tuple(x.shape[:-1]) + (n * C,), dtype=self.OP, device=x.device)
        assert out.is_contiguous() and out.numel() == n * x2.shape[0] * C and out.dtype == self.OP
        if not self.x3:   # "fp16": hi alone
            call("wf_cast_f16", x2.data_ptr(), x2.stride(0), out.data_ptr(), C, x2.shape[0], C, ops.stream())
            return out
        call("wf_split_f16x3" if self.f16 else "wf_split_bf16x3", x2.data_ptr(), x2.stride(0), out.data_ptr(), 3 * C, x2.shape[0], C, side, ops.stream())
        return out

    def _gemm(self, x, w, bias, out, epi, wkey=None):
        """out[M,N] = epi(x[M,K] @ w[N,K]^T + bias) on the operand type of this precision mode (wf_gemm_bf16 / wf_gemm_f16).  wkey: the weight
        dict key of `w` when it is a stored weight operand (its power-of-two scale is undone in the epilogue), None for activations."""
        if not self.f16:
            return gemm(x, w, bias, out, epi)
        M, K = x.shape
        N = w.shape[0]
        assert w.shape[1] == K and x.dtype == self.OP and w.dtype == self.OP and out.shape[0] == M and out.shape[1] == N
        call("wf_gemm_f16", x.data_ptr(), w.data_ptr(), bias.data_ptr() if bias is not None else None, out.data_ptr(), M, N, K,
             x.stride(0), w.stride(0), out.stride(0), epi, float(self.w.get(wkey + ".scale", 1.0)) if wkey else 1.0, ops.stream())
        return out

    def _note_range(self, what: str):
        """fp16 operand formats: a producer that met a value beyond +-65504 (or a NaN) raised the sticky device flag.  Queue a 4-byte
        copy of it into pinned host memory behind this call's kernels (wf_f16_overflow_flag_async) and an event -- NO host
        synchronisation inside encode / decode (no hidden device syncs; round 4 synchronised the stream once per VAE call).
        `check_range` reads it, DETERMINISTICALLY: every encode / decode / decode_blend_encode starts by waiting for the flag
        of the call before it (by then that call's kernels have all but finished: the wait costs a launch latency, not a drain) -- on a
        sharded VAE too, where the calls are collective and every rank is at the same program point -- and the schedulers' `fuse_latents`
        and the pipelines check at their own synchronisation points (FLF gate read-back, final frames).  What remains for the caller: the
        LAST call of a job.  A caller of the bare encode / decode API must call check_range() before consuming the result, or construct the
        VAE with strict_range=True (the check then runs inside the call, one host synchronisation each)."""
        if not self.f16:
            return
        self.check_range()   # an earlier call's flag (none, when the entry point ran its own check at the start)
        if getattr(self, "_flag_host", None) is None:
            self._flag_host = torch.zeros(1, dtype=torch.int32).pin_memory()
        call("wf_f16_overflow_flag_async", self._flag_host.data_ptr(), ops.stream())
        ev = torch.cuda.Event()
        ev.record()
        self._flag_pending = (ev, what)
        if self.strict_range:
            self.check_range()

    def check_range(self, wait: bool = True):
        """Turn a raised fp16 range flag of an earlier encode / decode into a RuntimeError -- never a silent inf.  wait=False: only if its
        copy has already landed.  With `comm` every rank takes part in one tiny all-gather of the flag first, so that all ranks fail
        together instead of one raising while its peers wait in the next collective; call it at the same program points on
        every rank (the pipelines do)."""
        pend = getattr(self, "_flag_pending", None)
        if pend is None:
            return
        ev, what = pend
        shared = self.comm is not None and getattr(self.comm, "world", 1) > 1
        if not wait and (shared or not ev.query()):
            return   # (wait=False is timing-dependent: a sharded VAE never takes it, every rank must take the same branch)
        ev.synchronize()
        self._flag_pending = None
        bad = int(self._flag_host[0])
        if shared:
            mine = torch.tensor([bad], dtype=torch.int32, device=self.device)
            allr = torch.empty((self.comm.world, 1), dtype=torch.int32, device=self.device)
            self.comm.all_gather(allr, mine)
            bad = int(allr.max())
        if bad:
            import ctypes
            flag = ctypes.c_int(0)
            call("wf_f16_overflow_flag", ctypes.byref(flag), 1, ops.stream())   # reset (synchronous: we are about to raise)
            raise RuntimeError(f"AutoencoderKLWan.{what}: an activation left the fp16 range (|x| > 65504) or was NaN under precision="
                               f"{self.precision!r}; use precision='bf16x3' (8-bit exponent) for these weights / inputs")

    def _res(self, x, p, cin, cout):
        """vae.py:186-220."""
        T, H, Wd, _ = x.shape
        W = self.w
        a = self._rms(x, W[p + ".residual.0.gamma"], blocked=True)
        y, _ = self._conv(a, p + ".residual.2", T, H, Wd, cout, (3, 3, 3), pt=2, ps=1)
        del a
        a2 = self._rms(y, W[p + ".residual.3.gamma"], blocked=True)
        del y
        if cin != cout:
            xb = self._operand(x)
            h = torch.empty((T, H, Wd, cout), dtype=F32, device=x.device)
            self._gemm(xb.view(-1, xb.shape[-1]), W[p + ".shortcut.w"], W[p + ".shortcut.b"], h.view(-1, cout), EPI_F32, wkey=p + ".shortcut.w")
            self.flops_last += 2 * T * H * Wd * xb.shape[-1] * cout
            del xb
        else:
            h = x
        out, _ = self._conv(a2, p + ".residual.6", T, H, Wd, cout, (3, 3, 3), pt=2, ps=1, resid=h)
        return out

    def _attn(self, x, p, rows=None):
        """vae.py:223-262: per-frame single-head attention over the H*W positions (C = 384).  rows = (r0, r1): only the query rows r0 .. r1
        of every frame are computed (keys / values: the whole frame) and the updated rows are returned as a slab [T, r1 - r0, W, C] -- the
        row-group sharded low-resolution stage; per output element the arithmetic is that of the whole-frame call."""
        T, H, Wd, C = x.shape
        hw = H * Wd
        # K / N padding for the MFMA GEMMs; padded score columns are never read by the softmax and are written as zero probabilities.  The
        # fp32-class path pads to 64 so that the batched P . V products (K = 3 hwp) qualify for the ping-pong GEMM (whole 64-wide K tiles)
        hwp = (hw + 63) // 64 * 64 if self.wide else (hw + 7) // 8 * 8
        W = self.w
        a = self._rms(x, W[p + ".norm.gamma"], silu=False)
        q0, q1 = (0, hw) if rows is None else (rows[0] * Wd, rows[1] * Wd)
        nq = q1 - q0
        if rows is not None:
            x = x[:, rows[0]:rows[1]].contiguous()
        if self.wide:
            return self._attn_x3(x, p, a, hw, hwp, q0, nq)
        qkv = torch.empty((T * hw + 8, 3 * C), dtype=BF, device=x.device)  # +8 rows: the padded K rows stay in-bounds
        qkv[T * hw:].zero_()
        self._gemm(a.view(-1, C), W[p + ".to_qkv.w"], W[p + ".to_qkv.b"], qkv[:T * hw], EPI_BF16)
        del a
        O = torch.empty((T * nq, C), dtype=BF, device=x.device)
        scale = 1.0 / math.sqrt(C)
        # all frames of a chunk in ONE launch per product (round 6; see _attn_x3): q / k are read in place from the qkv rows (row stride 3C,
        # frame stride hw rows), per frame the tiles and the K order of the single call
        tb = max(1, min(T, self.ATTN_BATCH_BYTES // (nq * hwp * 6 + C * hwp * 2)))
        S = torch.empty((tb, nq, hwp), dtype=F32, device=x.device)
        P = torch.empty((tb, nq, hwp), dtype=BF, device=x.device)
        Vt = torch.empty((tb, C, hwp), dtype=BF, device=x.device)
        for t0 in range(0, T, tb):
            n = min(tb, T - t0)
            call("wf_gemm_bf16_batched", qkv[t0 * hw + q0:].data_ptr(), qkv[t0 * hw:, C:].data_ptr(), S.data_ptr(), n, nq, hwp, C, 3 * C, 3 * C, hwp,
                 hw * 3 * C, hw * 3 * C, nq * hwp, EPI_F32, ops.stream())
            call("wf_softmax_rows", S.data_ptr(), hwp, P.data_ptr(), hwp, n * nq, hw, float(scale), ops.stream())
            for t in range(t0, t0 + n):
                call("wf_transpose_bf16", qkv[t * hw:, 2 * C:].data_ptr(), 3 * C, Vt[t - t0].data_ptr(), hwp, hw, C, ops.stream())
            call("wf_gemm_bf16_batched", P.data_ptr(), Vt.data_ptr(), O[t0 * nq:].data_ptr(), n, nq, C, hwp, hwp, hwp, C, nq * hwp, C * hwp, nq * C,
                 EPI_BF16, ops.stream())
        self._gemm(O, W[p + ".proj.w"], W[p + ".proj.b"], x.view(-1, C), EPI_F32_ACC)  # x + proj(attn)  (vae.py:262)
        self.flops_last += T * (4 * nq * hw * C) + 2 * T * hw * C * 3 * C + 2 * T * nq * C * C
        return x

    def _attn_x3(self, x, p, a, hw, hwp, q0, nq):
        """_attn with fp32-class contractions: q / k / v, the scores and the probabilities stay f32 and are split per use.  x: the rows that
        are updated ([T, rows, W, C]: the whole frames, or the query-row slab); a: the normalised WHOLE frames.  Also the one-term fp16
        mode (n = 1 operand term instead of 3): the f32 tensors between the products are the same, each operand is hi alone."""
        T, C = x.shape[0], x.shape[-1]
        W = self.w
        n3 = self.terms
        qkv = torch.zeros((T * hw + 64, 3 * C), dtype=F32, device=x.device)  # + 64 rows: the last frame's padded K rows stay in-bounds
        self._gemm(a.view(-1, n3 * C), W[p + ".to_qkv.w"], W[p + ".to_qkv.b"], qkv[:T * hw], EPI_F32, wkey=p + ".to_qkv.w")
        del a
        Of = torch.empty((T * nq, C), dtype=F32, device=x.device)
        scale = 1.0 / math.sqrt(C)
        # A frame's products are a fraction of the chip (Q . K^T: nq x hwp x n3 C = 175 workgroups of the 256 x 256 kernel at 480p, 49 on a row
        # group's query slab; P . V: 147 / 37), 21 times in a row, with a handful of small launches each.  Round 4 batched the P . V products
        # of a chunk of frames into ONE launch; round 6 does the same for Q . K^T, the softmax and the operand splits: q and k of ALL frames
        # are split once (two launches), the batched GEMM addresses frame t's rows in place (batch stride hw rows; the padded key rows of a
        # frame are the next frame's first rows, never read by the softmax), one softmax launch covers the chunk's n * nq rows, one split
        # each its probabilities and its transposed values.  Per frame the arithmetic is that of the single call (same kernels, same tiles,
        # same K order: the GEMM's results do not depend on the tile or on the batch).  The chunk is bounded by a byte budget
        # (ATTN_BATCH_BYTES: 11.8 GB at 480p = all 21 frames; 720p: 4 frames at a time): whole frames while they fit, never fewer than one.
        Q3 = self._operand(qkv[:, 0:C], 0)          # [T hw + 64, n3 C]
        K3 = self._operand(qkv[:, C:2 * C], 1)
        ld3 = n3 * C
        per_frame = nq * hwp * 8 + (nq + C) * n3 * hwp * 2 + C * hwp * 4
        tb = max(1, min(T, self.ATTN_BATCH_BYTES // per_frame))
        S = torch.empty((tb, nq, hwp), dtype=F32, device=x.device)
        P = torch.empty((tb, nq, hwp), dtype=F32, device=x.device)
        Vt = torch.empty((tb, C, hwp), dtype=F32, device=x.device)
        P3 = torch.empty((tb, nq, n3 * hwp), dtype=self.OP, device=x.device)
        V3 = torch.empty((tb, C, n3 * hwp), dtype=self.OP, device=x.device)
        batched = "wf_gemm_f16_batched" if self.f16 else "wf_gemm_bf16_batched"
        for t0 in range(0, T, tb):
            n = min(tb, T - t0)
            call(batched, Q3[t0 * hw + q0:].data_ptr(), K3[t0 * hw:].data_ptr(), S.data_ptr(), n, nq, hwp, ld3, ld3, ld3, hwp, hw * ld3, hw * ld3,
                 nq * hwp, EPI_F32, ops.stream())
            call("wf_softmax_rows_f32", S.data_ptr(), hwp, P.data_ptr(), hwp, n * nq, hw, float(scale), ops.stream())
            for t in range(t0, t0 + n):
                call("wf_transpose_f32", qkv[t * hw:, 2 * C:].data_ptr(), 3 * C, Vt[t - t0].data_ptr(), hwp, hw, C, ops.stream())
            self._operand(P[:n].view(n * nq, hwp), 0, out=P3[:n].view(n * nq, n3 * hwp))
            self._operand(Vt[:n].view(n * C, hwp), 1, out=V3[:n].view(n * C, n3 * hwp))
            call(batched, P3.data_ptr(), V3.data_ptr(), Of[t0 * nq:].data_ptr(), n, nq, C,
                 n3 * hwp, n3 * hwp, n3 * hwp, C, nq * n3 * hwp, C * n3 * hwp, nq * C, EPI_F32, ops.stream())
        del Q3, K3, S, P, Vt
        del P3, V3
        self._gemm(self._operand(Of), W[p + ".proj.w"], W[p + ".proj.b"], x.view(-1, C), EPI_F32_ACC, wkey=p + ".proj.w")
        self.flops_last += n3 * (T * (4 * nq * hw * C) + 2 * T * hw * C * 3 * C + 2 * T * nq * C * C)
        return x

    def _down(self, x, p, C, temporal):
        """vae.py:87-96, 139-159."""
        T, H, Wd, _ = x.shape
        Ho, Wo = H // 2, Wd // 2
        xb = self._operand(x)
        if not temporal or T == 1:
            y, _ = self._conv(xb, p + ".resample.1", T, Ho, Wo, C, (1, 3, 3), ss=2, ps=0)
            return y
        y, yb = self._conv(xb, p + ".resample.1", T, Ho, Wo, C, (1, 3, 3), ss=2, ps=0, out_f32=True, out_bf16=not self.wide)
        del xb
        if self.wide:
            yb = self._operand(y)
        To = (T - 1) // 2
        out = torch.empty((1 + To, Ho, Wo, C), dtype=F32, device=x.device)
        out[0].copy_(y[0])  # frame 0 by-passes time_conv (vae.py:146-148)
        W = self.w
        call("wf_conv3d_cl" + self._sfx, yb.data_ptr(), W[p + ".time_conv.w"].data_ptr(), W[p + ".time_conv.b"].data_ptr(), None,
             out[1:].data_ptr(), None, T, Ho, Wo, yb.shape[-1], To, Ho, Wo, C, 3, 1, 1, 2, 1, 0, 0, 0, 0, 0, None, *self._acc_scale(p + ".time_conv.w"), ops.stream())
        self.flops_last += 2 * To * Ho * Wo * C * 3 * yb.shape[-1]
        return out

    def _up(self, x, p, C, temporal):
        """vae.py:76-86, 101-141."""
        T, H, Wd, _ = x.shape
        xb = self._time_up(self._operand(x), p, C) if temporal else self._operand(x)
        Tn = xb.shape[0]
        if p + ".resample.1.ph00.w" in self.w:
            return self._up_phases(xb, p + ".resample.1", Tn, H, 2 * H, 0, C // 2)
        out, _ = self._conv(xb, p + ".resample.1", Tn, 2 * H, 2 * Wd, C // 2, (1, 3, 3), ps=1, up2=True)
        return out

    def _up_phases(self, xb, p, T, n_rows, out_rows, row_off, Cout):
        """The four phase convolutions of an upsampling conv.  xb [T, Hs, W, Cin] bf16: the whole image (Hs = n_rows, row_off = 0) or a row
        slab whose first row is source row s0 while the first output row belongs to source row y0 / 2 (row_off = s0 - y0 / 2 <= 0: halo rows
        above).  Writes out [T, out_rows, 2W, Cout] fp32: phase (py, px) computes n_rows x W outputs from source rows {y - 1 + py, y + py}
        (top padding 1 - py + row_off) and scatters them to rows 2 j + py, columns 2 x + px."""
        _, Hs, Wd, Cin = xb.shape
        out = torch.empty((T, out_rows, 2 * Wd, Cout), dtype=F32, device=xb.device)
        W = self.w
        for py in range(2):
            for px in range(2):
                call("wf_conv3d_cl_scatter" + self._sfx, xb.data_ptr(), W[f"{p}.ph{py}{px}.w"].data_ptr(), W[p + ".b"].data_ptr(), None, out.data_ptr(), None,
                     T, Hs, Wd, Cin, T, n_rows, Wd, Cout, 1, 2, 2, 1, 1, 0, 1 - py + row_off, 1 - px, self._zero_page().data_ptr(),
                     out_rows, 2 * Wd, 2, py, 2, px, *self._acc_scale(f"{p}.ph{py}{px}.w"), ops.stream())
        self.flops_last += 4 * 2 * T * n_rows * Wd * Cout * 4 * Cin
        return out

    def _run(self, x, plan):
        W = self.w
        for kind, p, cin, cout in plan:
            T, H, Wd, _ = x.shape
            if kind == "conv_in":  # x arrives bf16, zero-padded to 32 channels
                x, _ = self._conv(x, p, T, H, Wd, cout, (3, 3, 3), pt=2, ps=1)
            elif kind == "res":
                x = self._res(x, p, cin, cout)
            elif kind == "attn":
                x = self._attn(x, p)
            elif kind in ("down2d", "down3d"):
                x = self._down(x, p, cin, kind == "down3d")
            elif kind in ("up2d", "up3d"):
                x = self._up(x, p, cin, kind == "up3d")
            elif kind == "head":
                a = self._rms(x, W[p + ".0.gamma"], blocked=True)
                x, _ = self._conv(a, p + ".2", T, H, Wd, (cout + 31) // 32 * 32, (3, 3, 3), pt=2, ps=1)
        return x

    # ------------------------------------------------------------------------------------------------------------
    # multi-GPU: row-slab sharding of the high-resolution stages.  The low-resolution stage (60 x 104 at 480p:
    # ~4 % of the FLOPs, and the only place with global spatial coupling -- the mid-block attention) is computed replicated;
    # every other layer runs on H/P rows per rank.  A 3x3 convolution needs one halo row from each neighbour: the ranks
    # all-gather their top / bottom rows of the convolution INPUT (bf16) and pick their neighbours' (zeros at the image edge).
    # Only all-gather is used.  Results are bit-identical to the unsharded path (same per-pixel accumulation order).
    # ------------------------------------------------------------------------------------------------------------
    def _halo_pad(self, a):
        """a [T,Hs,W,C] (or slice-major [T,Hs,S,W,16]) bf16 -> [T,Hs+2,...] with the neighbours' boundary rows (zeros at the image
        boundary)."""
        T, Hs = a.shape[:2]
        out = torch.empty((T, Hs + 2) + tuple(a.shape[2:]), dtype=a.dtype, device=a.device)
        out[:, 1:Hs + 1].copy_(a)
        return self._halo_fill(out)

    def _halo_pad_of(self, x):
        """The conv operand of the row slab x f32 [T,Hs,W,C], halo-padded: [T,Hs+2,W,..] with the neighbours' boundary rows (zeros at the image
        edge).  The two border rows are produced FIRST: their operand rows travel to the neighbours on the communication stream
        (Comm.neighbor_rows_async) while the whole slab's operand is produced -- straight into rows 1 .. Hs of the padded buffer
        (wf_operand_rows: round 5 produced the slab's operand and copied it in, 1-2 GB per full-resolution layer).  Per pixel the same
        arithmetic wherever the pixel sits: bit-identical to exchanging rows of the produced slab."""
        T, Hs, Wd, C = x.shape
        e_op = self._operand(torch.stack([x[:, 0], x[:, Hs - 1]], dim=1))
        pending = self.comm.neighbor_rows_async(e_op[:, 0], e_op[:, 1], self._reps)
        nC = self.terms * C
        out = torch.empty((T, Hs + 2, Wd, nC), dtype=self.OP, device=x.device)
        fmt = (1 if self.f16 else 0) if self.x3 else (2 if self.f16 else 3)
        call("wf_operand_rows", x.data_ptr(), C, out.data_ptr(), nC, T * Hs * Wd, C, fmt, 0, Hs * Wd, 2 * Wd, Wd, ops.stream())
        up, down = pending()
        if up is not None:
            out[:, 0].copy_(up)
        else:
            out[:, 0].zero_()
        if down is not None:
            out[:, Hs + 1].copy_(down)
        else:
            out[:, Hs + 1].zero_()
        return out

    def _halo_fill(self, out):
        """out [T,Hs+2,...] whose rows 1 .. Hs are this rank's: rows 0 and Hs+1 <- the neighbours' boundary rows (zeros at the image edge).
        Round 5: exchanged with the two neighbours only (parallel.Comm.neighbor_rows: all-gathers inside two-rank groups), not all-gathered
        over the whole job -- a rank receives the 2 rows it needs instead of 2 (P - 1)."""
        comm = self.comm
        T, Hs = out.shape[0], out.shape[1] - 2
        a = out[:, 1:Hs + 1]
        reps = self._reps  # ranks per row group (1: every rank its own slab; > 1: `reps` consecutive ranks hold the same slab)
        up, down = comm.neighbor_rows(a[:, 0], a[:, Hs - 1], reps)
        if up is not None:
            out[:, 0].copy_(up)
        else:
            out[:, 0].zero_()
        if down is not None:
            out[:, Hs + 1].copy_(down)
        else:
            out[:, Hs + 1].zero_()
        return out

    def _halo_operand(self, x, gamma):
        """RMS_norm + SiLU of a row slab as the halo-padded conv operand: the norm kernel writes the slab's own rows in place (no copy)."""
        C = x.shape[-1]
        if C % 32 == 0:
            # the two border rows first: their operand rows travel to the neighbours on the communication stream while the norm kernel
            # produces the whole slab (per pixel the same arithmetic wherever the pixel sits: bit-identical to exchanging rows of the slab)
            T, Hs = x.shape[:2]
            edge = torch.stack([x[:, 0], x[:, Hs - 1]], dim=1)                       # [T, 2, W, C] f32
            e_op = self._rms(edge, gamma, blocked=True)                              # [T, 2, S, W, 16]
            pending = self.comm.neighbor_rows_async(e_op[:, 0], e_op[:, 1], self._reps)
            out = self._rms(x, gamma, blocked=True, halo=True)
            up, down = pending()
            if up is not None:
                out[:, 0].copy_(up)
            else:
                out[:, 0].zero_()
            if down is not None:
                out[:, Hs + 1].copy_(down)
            else:
                out[:, Hs + 1].zero_()
            return out
        return self._halo_pad(self._rms(x, gamma, blocked=True))

    def _rows_from_full(self, x, r0, r1):
        """Rows [r0, r1) of a replicated [T,H,W,C] tensor with zero rows outside the image -> contiguous slab."""
        T, H, Wd, C = x.shape
        out = torch.zeros((T, r1 - r0, Wd, C), dtype=x.dtype, device=x.device)
        lo, hi = max(r0, 0), min(r1, H)
        out[:, lo - r0:hi - r0].copy_(x[:, lo:hi])
        return out

    def _gather_rows(self, slab):
        """[T,Hs,W,C] per rank -> full [T,P*Hs,W,C] on every rank."""
        comm = self.comm
        allr = torch.empty((comm.world,) + tuple(slab.shape), dtype=slab.dtype, device=slab.device)
        comm.all_gather(allr, slab.contiguous())
        if self._reps > 1:  # one copy per row group
            allr = allr[::self._reps]
        return allr.permute(1, 0, 2, 3, 4).reshape(slab.shape[0], allr.shape[0] * slab.shape[1], slab.shape[2], slab.shape[3]).contiguous()

    def _attn_group(self, x, p, g: int, Hg: int):
        """The mid-block attention of row group g (x: the group's slab [T, Hg, W, C], the same on the group's `reps` ranks): keys / values are
        the whole frame (gathered), and the group's QUERY rows are divided among its replicas (round 6: each of the `reps` ranks computed all
        Hg rows before -- at 8 ranks, 4 groups x 2, every rank did a quarter of the queries instead of an eighth); the replicas' rows come back
        by one all-gather.  Per output row the arithmetic of the whole-frame call: bit-identical."""
        comm, reps = self.comm, self._reps
        full = self._gather_rows(x)
        if reps == 1:
            return self._attn(full, p, rows=(g * Hg, (g + 1) * Hg))
        T, _, Wd, C = x.shape
        j, per = comm.rank % reps, -(-Hg // reps)
        r0, r1 = min(j * per, Hg), min((j + 1) * per, Hg)
        mine = torch.zeros((T, per, Wd, C), dtype=F32, device=x.device)
        if r1 > r0:
            mine[:, :r1 - r0].copy_(self._attn(full, p, rows=(g * Hg + r0, g * Hg + r1)))
        allr = torch.empty((comm.world,) + tuple(mine.shape), dtype=F32, device=x.device)
        comm.all_gather(allr, mine)
        parts = []
        for jj in range(reps):
            n = min((jj + 1) * per, Hg) - min(jj * per, Hg)
            if n > 0:
                parts.append(allr[g * reps + jj, :, :n])
        return torch.cat(parts, dim=1).contiguous()

    def _row_groups(self, h: int, even: bool = False) -> int:
        """Row groups for a stage of h rows: the largest divisor G of the world size with h % G == 0 (and an even number of rows per group
        if a stride-2 conv follows); world / G consecutive ranks then compute the same slab (the low-resolution stage: 60 rows on 8 ranks =
        4 groups of 15, computed twice each, instead of the whole stage on every rank)."""
        P = self.comm.world
        for G in range(P, 0, -1):
            if P % G == 0 and h % G == 0 and (not even or (h // G) % 2 == 0):
                return G
        return 1

    def _res_slab(self, x, p, cin, cout):
        T, Hs, Wd, _ = x.shape
        W = self.w
        a = self._halo_operand(x, W[p + ".residual.0.gamma"])
        y, _ = self._conv(a, p + ".residual.2", T, Hs, Wd, cout, (3, 3, 3), pt=2, ps=1, ph=0)
        del a
        a2 = self._halo_operand(y, W[p + ".residual.3.gamma"])
        del y
        if cin != cout:
            xb = self._operand(x)
            h = torch.empty((T, Hs, Wd, cout), dtype=F32, device=x.device)
            self._gemm(xb.view(-1, xb.shape[-1]), W[p + ".shortcut.w"], W[p + ".shortcut.b"], h.view(-1, cout), EPI_F32, wkey=p + ".shortcut.w")
            del xb
        else:
            h = x
        out, _ = self._conv(a2, p + ".residual.6", T, Hs, Wd, cout, (3, 3, 3), pt=2, ps=1, ph=0, resid=h)
        return out

    def _time_up(self, xb, p, C):
        """'upsample3d' temporal part (pointwise in space) on the conv operand xb of x: [T,..] -> [1+2(T-1),..] as an operand again.
        The first latent frame by-passes time_conv ('Rep', vae.py:106-108): its operand is carried over as it is.  precision="fp32":
        the interleaved frames are produced in f32 and split."""
        T, H, Wd, Cop = xb.shape
        if T == 1:
            return xb
        W = self.w
        self.flops_last += 2 * (T - 1) * H * Wd * 2 * C * 3 * Cop
        if self.wide:
            yf = torch.empty((1 + 2 * (T - 1), H, Wd, C), dtype=F32, device=xb.device)  # frame 0 is not written (tsplit: 1 + 2t + h)
            call("wf_conv3d_cl" + self._sfx, xb[1:].data_ptr(), W[p + ".time_conv.w"].data_ptr(), W[p + ".time_conv.b"].data_ptr(), None,
                 yf.data_ptr(), None, T - 1, H, Wd, Cop, T - 1, H, Wd, 2 * C, 3, 1, 1, 1, 1, 2, 0, 0, 0, 1, None, *self._acc_scale(p + ".time_conv.w"), ops.stream())
            yb = torch.empty((1 + 2 * (T - 1), H, Wd, Cop), dtype=self.OP, device=xb.device)
            yb[0].copy_(xb[0])
            self._operand(yf[1:], out=yb[1:])
            return yb
        yb = torch.empty((1 + 2 * (T - 1), H, Wd, C), dtype=BF, device=xb.device)
        yb[0].copy_(xb[0])
        call("wf_conv3d_cl" + self._sfx, xb[1:].data_ptr(), W[p + ".time_conv.w"].data_ptr(), W[p + ".time_conv.b"].data_ptr(), None,
             None, yb.data_ptr(), T - 1, H, Wd, C, T - 1, H, Wd, 2 * C, 3, 1, 1, 1, 1, 2, 0, 0, 0, 1, None, *self._acc_scale(p + ".time_conv.w"), ops.stream())
        return yb

    def _up_slab(self, xsrc_b, p, C, temporal, s0, y0, Ho, h_src):
        """Nearest-2x upsample + 3x3 conv for output rows [y0, y0+Ho) (upsampled coordinates) from a bf16 source slab that
        starts at source row s0 (may be -1: zero row) and covers (y0-1)>>1 .. (y0+Ho)>>1.  Output row j reads upsampled rows
        y0 + j - 1 + dy, i.e. slab-local upsampled row j + dy - ph with ph = 1 + 2*s0 - y0."""
        if temporal:
            xsrc_b = self._time_up(xsrc_b, p, C)
            # rows outside the image are the spatial conv's ZERO padding: the (pointwise) time_conv turned them into its bias
            if s0 < 0:
                xsrc_b[:, 0].zero_()
            if s0 + xsrc_b.shape[1] > h_src:
                xsrc_b[:, -1].zero_()
        T, _, Wd, _ = xsrc_b.shape
        if p + ".resample.1.ph00.w" in self.w:
            # output rows y0 .. y0 + Ho (both even here): phase py computes source rows y0/2 .. y0/2 + Ho/2, reading slab rows
            # (y + a - 1 + py) - s0 for its two taps a: the slab starts y0/2 - s0 rows above the first of them
            assert y0 % 2 == 0 and Ho % 2 == 0
            return self._up_phases(xsrc_b, p + ".resample.1", T, Ho // 2, Ho, s0 - y0 // 2, C // 2)
        out, _ = self._conv(xsrc_b, p + ".resample.1", T, Ho, 2 * Wd, C // 2, (1, 3, 3), ps=1, ph=1 + 2 * s0 - y0, up2=True)
        return out

    def _down_slab(self, x, p, C, temporal):
        """fp32 slab [T,Hs,W,C] -> [T',Hs/2,W/2,C]  (ZeroPad2d((0,1,0,1)) + stride-2 conv: bottom halo row only)."""
        T, Hs, Wd, _ = x.shape
        xpad = self._halo_pad_of(x)
        Ho, Wo = Hs // 2, Wd // 2
        if not temporal or T == 1:
            y, _ = self._conv(xpad, p + ".resample.1", T, Ho, Wo, C, (1, 3, 3), ss=2, ps=0, ph=-1)
            return y
        y, yb = self._conv(xpad, p + ".resample.1", T, Ho, Wo, C, (1, 3, 3), ss=2, ps=0, ph=-1, out_f32=True, out_bf16=not self.wide)
        if self.wide:
            yb = self._operand(y)
        To = (T - 1) // 2
        out = torch.empty((1 + To, Ho, Wo, C), dtype=F32, device=x.device)
        out[0].copy_(y[0])
        W = self.w
        call("wf_conv3d_cl" + self._sfx, yb.data_ptr(), W[p + ".time_conv.w"].data_ptr(), W[p + ".time_conv.b"].data_ptr(), None,
             out[1:].data_ptr(), None, T, Ho, Wo, yb.shape[-1], To, Ho, Wo, C, 3, 1, 1, 2, 1, 0, 0, 0, 0, 0, None, *self._acc_scale(p + ".time_conv.w"), ops.stream())
        return out

    def can_shard(self, H_lat: int) -> bool:
        """Row sharding needs every sharded stage to hold an integer (even, where a stride-2 conv follows) number of rows."""
        if self.comm is None or self.comm.world == 1:
            return False
        P = self.comm.world
        return (2 * H_lat) % P == 0 and (8 * H_lat) % (4 * P) == 0

    def _decode_one_sharded(self, z: torch.Tensor, gather: bool = True, crop=None) -> torch.Tensor:
        """gather=False: return this rank's row slab [3, F, 8h/P, 8w] (rows rank * 8h/P ...) instead of the gathered video.  crop: the
        latent columns decoded after the latent-resolution stage (decode(columns=...))."""
        comm = self.comm
        P, rank = comm.world, comm.rank
        C, T, h, w = z.shape
        x = torch.empty((T, h, w, Z_DIM), dtype=F32, device=self.device)
        call("wf_ncthw_to_cl", z.data_ptr(), x.data_ptr(), None, Z_DIM, Z_DIM, T * h * w, ops.stream())
        x = self._latent_in(x, T, h, w)
        plan = decoder_plan()
        first_up = next(i for i, e in enumerate(plan) if e[0] in ("up2d", "up3d"))
        G = self._row_groups(h)
        if G > 1:
            # stage 0 (conv_in, 5 residual blocks, the mid-block attention at h x w) in G row groups of h / G rows, world / G ranks per group
            self._reps = P // G
            try:
                g, Hg = rank // self._reps, h // G
                kind, p, cin, cout = plan[0]
                x, _ = self._conv(self._rows_from_full(x, g * Hg - 1, (g + 1) * Hg + 1), p, T, Hg, w, cout, (3, 3, 3), pt=2, ps=1, ph=0)
                for kind, p, cin, cout in plan[1:first_up]:
                    if kind == "res":
                        x = self._res_slab(x, p, cin, cout)
                    else:  # attention: the keys / values are the whole frame, the queries this group's rows
                        x = self._attn_group(x, p, g, Hg)
                x = self._gather_rows(x)                       # whole frames on every rank again: [T,h,w,384] fp32
            finally:
                self._reps = 1
        else:
            x = self._run(x, plan[:first_up])                  # stage 0 replicated: [T,h,w,384] fp32
        if crop is not None:
            x = x[:, :, crop[0]:crop[1]].contiguous()
        kind, p, cin, cout = plan[first_up]
        Hs = 2 * h // P                                        # my rows at the next resolution
        y0 = rank * Hs
        s0, s1 = (y0 - 1) >> 1, ((y0 + Hs) >> 1) + 1               # source rows incl. halo (s0 = -1 -> zero row)
        src = self._operand(self._rows_from_full(x, s0, s1))       # (rows first: the operand of this rank's rows only; zero rows stay zero)
        x = self._up_slab(src, p, cin, kind == "up3d", s0, y0, Hs, h)
        h_cur = 2 * h                                          # full image height at the current resolution
        row0 = y0                                              # global first row of my slab at the current resolution
        for kind, p, cin, cout in plan[first_up + 1:]:
            if kind == "res":
                x = self._res_slab(x, p, cin, cout)
            elif kind in ("up2d", "up3d"):
                Hcur = x.shape[1]
                x = self._up_slab(self._halo_pad_of(x), p, cin, kind == "up3d", row0 - 1, 2 * row0, 2 * Hcur, h_cur)
                row0 *= 2
                h_cur *= 2
            elif kind == "head":
                Tn, Hn, Wn, _ = x.shape
                a = self._halo_operand(x, self.w[p + ".0.gamma"])
                x, _ = self._conv(a, p + ".2", Tn, Hn, Wn, (cout + 31) // 32 * 32, (3, 3, 3), pt=2, ps=1, ph=0)
        # channels-last (32 padded channels) -> [3, F, rows, W] on the SLAB, then -- for the gathered form -- one all-gather of the 3-channel
        # slabs (round 5 gathered the padded 32-channel slabs, 4.1 GB at C2, and converted the whole video on every rank)
        x = x.contiguous()
        Fo, Hs_, Wo, Cy = x.shape
        out = torch.empty((3, Fo, Hs_, Wo), dtype=F32, device=self.device)
        call("wf_cl_to_ncthw", x.data_ptr(), out.data_ptr(), 3, Cy, Fo * Hs_ * Wo, 1.0, ops.stream())
        if gather:
            allr = torch.empty((P, 3, Fo, Hs_, Wo), dtype=F32, device=self.device)
            comm.all_gather(allr, out)
            out = allr.permute(1, 2, 0, 3, 4).reshape(3, Fo, P * Hs_, Wo).contiguous()
        return out if crop is None else self._uncrop(out, crop, w)

    def _encode_one_sharded(self, video: torch.Tensor, slab: torch.Tensor = None) -> torch.Tensor:
        """video [3,F,H,W] replicated on every rank -- or slab [3,F,H/P,W]: only this rank's rows (rank * H/P ...), the halo rows of the
        first convolution then come from the neighbours by all-gather like every later layer's (same operand values: bit-identical)."""
        comm = self.comm
        P, rank = comm.world, comm.rank
        plan = encoder_plan()
        if slab is not None:
            C, Fr, Hs, Wd = slab.shape
            H = Hs * P
            xpad = self._halo_pad(self._video_in(slab))
        else:
            C, Fr, H, Wd = video.shape
            Hs = H // P
            y0 = rank * Hs
            lo, hi = max(y0 - 1, 0), min(y0 + Hs + 1, H)      # this rank's rows + halo: only those are converted (round 5: the whole video)
            xs = self._video_in(video[:, :, lo:hi].contiguous())
            xpad = self._rows_from_full(xs, y0 - 1 - lo, y0 + Hs + 1 - lo)
        kind, p, cin, cout = plan[0]
        x, _ = self._conv(xpad, p, Fr, Hs, Wd, cout, (3, 3, 3), pt=2, ps=1, ph=0)
        downs = [i for i, e in enumerate(plan) if e[0] in ("down2d", "down3d")]
        last_down = downs[-1]
        for kind, p, cin, cout in plan[1:last_down]:
            if kind == "res":
                x = self._res_slab(x, p, cin, cout)
            else:
                x = self._down_slab(x, p, cin, kind == "down3d")
        x = self._gather_rows(x)                                # whole frames at the resolution of the last downsample
        G = self._row_groups(x.shape[1], even=True)
        if G > 1:
            # the last downsample and everything after it (4 residual blocks, the mid-block attention, the head) in G row groups
            self._reps = P // G
            try:
                g, Hg = rank // self._reps, x.shape[1] // G
                x = x[:, g * Hg:(g + 1) * Hg].contiguous()
                for kind, p, cin, cout in plan[last_down:]:
                    if kind == "res":
                        x = self._res_slab(x, p, cin, cout)
                    elif kind in ("down2d", "down3d"):
                        x = self._down_slab(x, p, cin, kind == "down3d")
                    elif kind == "attn":
                        x = self._attn_group(x, p, g, x.shape[1])
                    else:  # head
                        Tn, Hn, Wn, _ = x.shape
                        a = self._halo_operand(x, self.w[p + ".0.gamma"])
                        x, _ = self._conv(a, p + ".2", Tn, Hn, Wn, (cout + 31) // 32 * 32, (3, 3, 3), pt=2, ps=1, ph=0)
                y = self._gather_rows(x)
            finally:
                self._reps = 1
        else:
            y = self._run(x, plan[last_down:])
        T, h, w, _ = y.shape
        q = self._small_conv(y, "conv1", T, h, w, 2 * Z_DIM, (1, 1, 1))
        out = torch.empty((2 * Z_DIM, T, h, w), dtype=F32, device=self.device)
        call("wf_cl_to_ncthw", q.data_ptr(), out.data_ptr(), 2 * Z_DIM, 2 * Z_DIM, T * h * w, 0.0, ops.stream())
        return out  # [mean | logvar]

    def _video_in(self, video):
        """[3,F,H,W] f32 -> channels-last conv operand [F,H,W,32 (x3)]: 3 channels zero-padded to one MFMA K slice."""
        C, Fr, H, Wd = video.shape
        if self.f16 and not self.x3:   # one-term fp16: the converting layout kernel raises the range flag itself
            x = torch.empty((Fr, H, Wd, 32), dtype=self.OP, device=self.device)
            call("wf_ncthw_to_cl_f16", video.data_ptr(), None, x.data_ptr(), 3, 32, Fr * H * Wd, ops.stream())
            return x
        if self.x3:
            xf = torch.empty((Fr, H, Wd, 32), dtype=F32, device=self.device)
            call("wf_ncthw_to_cl", video.data_ptr(), xf.data_ptr(), None, 3, 32, Fr * H * Wd, ops.stream())
            return self._operand(xf)
        x = torch.empty((Fr, H, Wd, 32), dtype=BF, device=self.device)
        call("wf_ncthw_to_cl", video.data_ptr(), None, x.data_ptr(), 3, 32, Fr * H * Wd, ops.stream())
        return x

    def _latent_in(self, x, T, h, w):
        """post-quant conv2 (vae.py:558) on the channels-last latent -> conv operand [T,h,w,32 (x3)] (16 channels + 16 zero channels)."""
        if self.wide:
            return self._operand(self._small_conv(x, "conv2", T, h, w, 32, (1, 1, 1), out_dtype=F32))
        return self._small_conv(x, "conv2", T, h, w, 32, (1, 1, 1), out_dtype=BF)

    # ------------------------------------------------------------------------------------------------------------
    # diffusers protocol
    # ------------------------------------------------------------------------------------------------------------
    def _encode_one(self, video: torch.Tensor) -> torch.Tensor:
        """[3,F,H,W] f32 -> posterior mean [16,T,h,w] f32  (vae.py:516-542 without the scale; mode() = mean half)."""
        C, Fr, H, Wd = video.shape
        if (Fr - 1) % 4 != 0:
            raise ValueError(f"number of frames must be 1 + 4k, got {Fr}")
        if H % 8 or Wd % 8:
            raise ValueError("height and width must be multiples of 8")
        self.flops_last = 0
        if self.can_shard(H // 8):
            return self._encode_one_sharded(video)
        x = self._video_in(video)
        y = self._run(x, encoder_plan())
        T, h, w, _ = y.shape
        q = self._small_conv(y, "conv1", T, h, w, 2 * Z_DIM, (1, 1, 1))
        out = torch.empty((2 * Z_DIM, T, h, w), dtype=F32, device=self.device)
        call("wf_cl_to_ncthw", q.data_ptr(), out.data_ptr(), 2 * Z_DIM, 2 * Z_DIM, T * h * w, 0.0, ops.stream())
        return out  # [mean | logvar]

    def _decode_one(self, z: torch.Tensor, crop=None) -> torch.Tensor:
        """[16,T,h,w] f32 -> [3, 4T-3, 8h, 8w] f32 clamped to [-1,1] (vae.py:544-568, autoencoder_kl_wan.py:1222).  crop = (a0, a1): see
        decode(columns=...)."""
        C, T, h, w = z.shape
        self.flops_last = 0
        if self.can_shard(h):
            return self._decode_one_sharded(z, crop=crop)
        x = torch.empty((T, h, w, Z_DIM), dtype=F32, device=self.device)
        call("wf_ncthw_to_cl", z.data_ptr(), x.data_ptr(), None, Z_DIM, Z_DIM, T * h * w, ops.stream())
        x = self._latent_in(x, T, h, w)  # 16 channels + 16 zero channels
        plan = decoder_plan()
        if crop is None:
            y = self._run(x, plan)
        else:
            first_up = next(i for i, e in enumerate(plan) if e[0] in ("up2d", "up3d"))
            x = self._run(x, plan[:first_up])                       # the latent-resolution stage sees whole frames (mid-block attention)
            y = self._run(x[:, :, crop[0]:crop[1]].contiguous(), plan[first_up:])
        Fo, Ho, Wo, Cy = y.shape
        out = torch.empty((3, Fo, Ho, Wo), dtype=F32, device=self.device)
        call("wf_cl_to_ncthw", y.data_ptr(), out.data_ptr(), 3, Cy, Fo * Ho * Wo, 1.0, ops.stream())
        return out if crop is None else self._uncrop(out, crop, w)

    # ------------------------------------------------------------------------------------------------------------
    # decoding only what an IRR injection consumes
    # ------------------------------------------------------------------------------------------------------------
    # The decoded video of fuse_latents (SCHED:1285) is consumed by ONE statement, fused = ref * m + dec * (1 - m) (SCHED:1380): wherever the
    # mask is exactly 1 the result is the reference pixel whatever (finite) value dec has there.  Real warping masks are mostly 1 (the
    # warped view covers most of the frame; the hole grows from one side: SURVEY 8d's synthetic mask ends at 65 % valid), so only the pixel
    # COLUMNS that hold a mask value != 1 -- plus the receptive field of the layers that follow -- need decoding.  Everything up to the first
    # up-sampling layer (4 % of the decoder's flops, and the only place with global spatial coupling: the mid-block attention) runs on whole
    # frames; the rest runs on the cropped columns.  Every needed pixel is computed by the same kernels on the same inputs inside its
    # receptive field, so the BLEND RESULT is bit-identical to decoding everything (tests/test_gpu_vae.py); the other columns of the
    # returned video are zeros.
    CROP_HALO = 8      # latent columns decoded beyond the needed ones on either side.  Receptive field of the layers after the crop, walking
                       # back from a needed output pixel: head conv 1 px + 3 residual blocks 6 = 7 @480 -> (+1 conv) / 2 = 4 @240 -> + 6 + 1 = 11
                       # -> 6 @120 -> + 6 + 1 = 13 -> 7 latent columns; 8 keeps the crop on the 64-pixel tile grid of the convolutions
    CROP_MIN_GAIN = 0.9  # crop only when it leaves at most this fraction of the columns

    def needed_columns(self, mask: torch.Tensor):
        """mask [1,1,F,H,W] f32 (aligned to the decoded size) -> (c0, c1): the LATENT columns whose pixels the blend can see (mask != 1
        somewhere in the column), or None when that is (nearly) everything.  One 8-byte read-back per distinct mask tensor (the mask of a job
        is the same object for all its injections)."""
        key = (mask.data_ptr(), mask._version, tuple(mask.shape))
        cache = self.__dict__.setdefault("_cols_cache", {})
        if key not in cache:
            cache.clear()
            W = mask.shape[-1]
            md = mask.to(device=self.device, dtype=F32).contiguous()
            out = torch.empty(2, dtype=torch.int32, device=self.device)
            call("wf_mask_column_range", md.data_ptr(), md.numel() // W, W, out.data_ptr(), ops.stream())
            x0, x1 = (int(v) for v in out.cpu())
            cache[key] = (mask, None if x1 <= x0 else (x0 // 8, (x1 + 7) // 8))   # (`mask` is held: its address cannot be recycled under the key)
        cols = cache[key][1]
        if cols is None:
            return (0, 0)   # no pixel of the decoded video reaches the result
        return cols

    def _crop_range(self, columns, w: int):
        """(c0, c1) needed latent columns -> the latent columns [a0, a1) to decode after the latent-resolution stage, or None (decode all)."""
        if columns is None:
            return None
        c0, c1 = columns
        if c1 <= c0:
            c0, c1 = 0, 8   # nothing is needed: the cheapest legal crop
        a0 = max(0, (c0 - self.CROP_HALO) // 8 * 8)
        a1 = min(w, -(-(c1 + self.CROP_HALO) // 8) * 8)
        return (a0, a1) if (a1 - a0) <= self.CROP_MIN_GAIN * w else None

    def _uncrop(self, out: torch.Tensor, crop, w: int) -> torch.Tensor:
        """[3, F, H, 8 (a1 - a0)] -> [3, F, H, 8 w] with zeros in the columns that were not decoded."""
        full = torch.zeros(tuple(out.shape[:-1]) + (8 * w,), dtype=out.dtype, device=out.device)
        full[..., 8 * crop[0]:8 * crop[1]].copy_(out)
        return full

    def _row_slab_of(self, t: torch.Tensor, y0: int, Hs: int) -> torch.Tensor:
        """Rows [y0, y0 + Hs) of a [1,C,F,H,W] tensor as a contiguous slab, kept while the tensor is unchanged (the reference video and the
        mask are the same objects for all 30 injections of a job)."""
        key = (t.data_ptr(), t._version, tuple(t.shape), y0, Hs)
        cache = self.__dict__.setdefault("_slab_cache", {})
        hit = cache.get(key)
        if hit is None:
            # two live entries (the reference video and the mask); an entry whose source differs from both is dropped, so a caller that
            # hands over fresh tensors every time pins at most two stale full-resolution sources, not four
            while len(cache) >= 2:
                cache.pop(next(iter(cache)))
            hit = cache[key] = (t, t[:, :, :, y0:y0 + Hs].contiguous())  # `t` is held so that its storage cannot be recycled under the key
        return hit[1]

    @torch.no_grad()
    def decode_blend_encode(self, z: torch.Tensor, ref: torch.Tensor, mask: torch.Tensor):
        """The pixel round trip of one IRR injection: decode (SCHED:1285) -> blend the warped reference in (SCHED:1375-1381) -> encode
        (SCHED:1384) -> the posterior.  z [1,16,T,h,w]; ref [1,3,F,H,W] / mask [1,1,F,H,W] fp32, already aligned to the decoded size.
        One GPU: exactly decode(), ops.blend_pixels(), encode().  Row-sharded (comm): the decoded video is NOT gathered -- the blend is
        element-wise per pixel, so each rank blends and re-encodes its own row slab (the encoder's first halo comes from the neighbours);
        only latent-resolution tensors are gathered.  Bit-identical to the gathered form."""
        self.check_range()   # the flag of the VAE call before this one (see _note_range)
        z = self._io_in(z)
        cols = self.needed_columns(mask) if self.crop_to_mask else None
        if z.shape[0] != 1 or not self.can_shard(z.shape[3]):
            dec = self.decode(z, return_dict=False, columns=cols)[0]
            return self.encode(ops.blend_pixels(ref, mask, dec)).latent_dist
        self.flops_last = 0
        slab = self._io_out(self._decode_one_sharded(z[0], gather=False, crop=self._crop_range(cols, z.shape[4])))  # [3, F, Hs, W] in the module dtype
        Hs = slab.shape[2]
        y0 = self.comm.rank * Hs
        fused = ops.blend_pixels(self._row_slab_of(ref, y0, Hs), self._row_slab_of(mask, y0, Hs), slab.unsqueeze(0))
        mom = self._io_out(self._encode_one_sharded(None, slab=self._io_in(fused)[0]).unsqueeze(0))
        self._note_range("decode_blend_encode")
        return _LatentDist(mom[:, :Z_DIM].contiguous(), mom[:, Z_DIM:])

    def _io_in(self, x: torch.Tensor) -> torch.Tensor:
        """A tensor entering the module: on the device, holding values of the module dtype (rounded to bf16 for a bf16 module: the
        reference's `.to(dtype=vae.dtype)`), as the contiguous f32 the kernels read."""
        x = x.to(self.device)
        if self.dtype == torch.bfloat16 and x.dtype != torch.bfloat16:
            x = ops.cast(x.contiguous(), torch.bfloat16)
        return x.contiguous() if x.dtype == F32 else ops.cast(x.contiguous(), F32)

    def _io_out(self, y: torch.Tensor) -> torch.Tensor:
        """A result leaving the module: in the module dtype."""
        return y if self.dtype == F32 else ops.cast(y.contiguous(), self.dtype)

    @torch.no_grad()
    def encode(self, x: torch.Tensor, return_dict: bool = True):
        """-> the posterior (latent_dist.mode() / .sample()).  fp16 operand modes: the range flag of THIS call is checked at the next VAE
        call or check_range() -- call check_range() before consuming the result if neither follows (or use strict_range=True)."""
        self.check_range()   # the flag of the VAE call before this one (see _note_range)
        x = self._io_in(x)
        moments = self._io_out(torch.stack([self._encode_one(v) for v in x]))
        self._note_range("encode")
        post = _LatentDist(moments[:, :Z_DIM].contiguous(), moments[:, Z_DIM:])
        if not return_dict:
            return (post,)
        return SimpleNamespace(latent_dist=post)

    @torch.no_grad()
    def decode(self, z: torch.Tensor, return_dict: bool = True, columns=None):
        """-> the video, clamped to [-1, 1].  fp16 operand modes: see encode() for when the range flag of this call is checked.
        columns = (c0, c1) (an extension of the diffusers protocol, used by the IRR injection only): the caller consumes the LATENT columns
        [c0, c1) of the video alone (pixel columns 8 c0 .. 8 c1, see needed_columns); those are what a full decode returns, bit for bit,
        the other columns are unspecified finite values (zeros, or decoded where they fall inside the halo)."""
        self.check_range()   # the flag of the VAE call before this one (see _note_range)
        z = self._io_in(z)
        crop = self._crop_range(columns, z.shape[4])
        out = self._io_out(torch.stack([self._decode_one(v, crop) for v in z]))
        self._note_range("decode")
        if not return_dict:
            return (out,)
        return SimpleNamespace(sample=out)
