"""Go / no-go for a Winograd F(2x2, 3x3) form of the VAE's 3 x 3 x 3 ResidualBlock convolutions (VERDICT r4 "next" #8: one calculation, no
kernel).  CPU only:  python tools/winograd_probe.py > profiles/r5_winograd_go_nogo.md

Part 1 (numerics, measured here): one 96 -> 96 and one 192 -> 192 causal 3 x 3 x 3 layer on RMS-norm + SiLU activations, weights drawn
like oracle/vae.random_weights; fp64 direct convolution = the reference; compared: (a) direct in the engine's fp16x3 arithmetic (hi / lo
fp16 parts, three product terms, fp32 accumulation, weights pre-scaled by a power of two as round 5 stores them), (b) Winograd F(2x2, 3x3)
over the two spatial axes with the SAME operand arithmetic applied to the TRANSFORMED operands (input transform B^T d B in fp32 before the
split, weight transform G g G^T in fp64 at load before the split, output transform A^T m A in fp32).
Part 2 (cost model, arithmetic on the shipped kernel's own numbers): accumulator footprint, operand fragments per MFMA, LDS / L2 traffic,
transform VALU on a lone wave.  See the table this prints.
"""
import math

import torch

torch.manual_seed(0)
F64, F32, F16 = torch.float64, torch.float32, torch.float16


def split3(x, scale_pow2=False):
    """-> (hi, lo, k): fp16 parts of x * 2^k (k = 0 unless scale_pow2: the per-matrix power-of-two scale of round 5's weight operands)."""
    k = 0
    if scale_pow2:
        k = 13 - math.floor(math.log2(float(x.abs().max())))
    xs = x.to(F32) * (2.0 ** k)
    hi = xs.to(F16)
    lo = (xs - hi.to(F32)).to(F16)
    return hi.to(F64), lo.to(F64), k


def contract3(a, w, eq):
    """fp16x3 contraction of activation a with weight w over einsum `eq`: hi.hi + lo.hi + hi.lo, products exact, accumulation in fp32 --
    emulated as fp64 products of the fp16 parts summed in fp64 and rounded to fp32 once (the fp32 accumulation error of ~sqrt(K) 2^-24 is
    the same for both forms and far below the split's)."""
    ah, al, _ = split3(a)
    wh, wl, k = split3(w, scale_pow2=True)
    acc = torch.einsum(eq, ah, wh) + torch.einsum(eq, al, wh) + torch.einsum(eq, ah, wl)
    return (acc * 2.0 ** -k).to(F32)


BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=F64)
G = torch.tensor([[1, 0, 0], [0.5, 0.5, 0.5], [0.5, -0.5, 0.5], [0, 0, 1]], dtype=F64)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=F64)


def layer(C, T=3, H=32, W=32):
    x = torch.randn(T, H, W, C, dtype=F64)
    x = x / x.pow(2).mean(-1, keepdim=True).sqrt()          # RMS-norm
    x = (x * torch.sigmoid(x)).to(F32)                       # SiLU -> the fp32 activation the producer splits
    w = (torch.randn(C, C, 3, 3, 3, dtype=F64) / math.sqrt(27 * C)).to(F32)   # [Cout, Cin, kt, kh, kw], oracle-like init
    xp = torch.nn.functional.pad(x.permute(3, 0, 1, 2).unsqueeze(0).to(F64), (1, 1, 1, 1, 2, 0))   # causal in time, zero-padded in space
    ref = torch.nn.functional.conv3d(xp, w.to(F64))[0].permute(1, 2, 3, 0)                            # [T, H, W, Cout] fp64
    # (a) direct, fp16x3: im2col patches [T, H, W, 27 * Cin] x [27 * Cin, Cout]
    cols = xp[0].unfold(1, 3, 1).unfold(2, 3, 1).unfold(3, 3, 1)           # [Cin, T, H, W, 3, 3, 3]
    cols = cols.permute(1, 2, 3, 0, 4, 5, 6).reshape(T, H, W, -1).to(F32)
    wm = w.reshape(C, -1)                                                    # [Cout, Cin * 27]
    direct = contract3(cols, wm, "thwk,ok->thwo")
    # (b) Winograd F(2x2, 3x3) on (h, w), temporal taps kept as an outer sum: 4 x 4 input tiles at stride 2
    tiles = xp[0].unfold(2, 4, 2).unfold(3, 4, 2)                           # [Cin, T + 2, H / 2, W / 2, 4, 4]
    V = torch.einsum("ij,cthwjk,lk->cthwil", BT, tiles, BT).to(F32)         # input transform, fp32 (adds only)
    U = torch.einsum("ij,ocdjk,lk->ocdil", G, w.to(F64), G).to(F32)         # weight transform at load: [Cout, Cin, kt, 4, 4]
    Mw = torch.zeros(T, H // 2, W // 2, 4, 4, C, dtype=F64)
    for xi in range(4):
        for nu in range(4):
            a = torch.stack([V[:, d:d + T, :, :, xi, nu] for d in range(3)], 0).permute(2, 3, 4, 0, 1).reshape(T, H // 2, W // 2, -1)   # [T, h, w, 3 * Cin]
            b = U[:, :, :, xi, nu].permute(0, 2, 1).reshape(C, -1)                                                                      # [Cout, 3 * Cin]
            Mw[:, :, :, xi, nu] = contract3(a, b, "thwk,ok->thwo").to(F64)
    Y = torch.einsum("ij,thwjko,lk->thwilo", AT, Mw.to(F32).to(F64), AT).to(F32)    # output transform in fp32
    wino = Y.permute(0, 1, 3, 2, 4, 5).reshape(T, H, W, C)
    rel = lambda y: float(((y.to(F64) - ref).pow(2).sum() / ref.pow(2).sum()).sqrt())   # noqa: E731
    mx = lambda y: float((y.to(F64) - ref).abs().max() / ref.abs().max())               # noqa: E731
    vmax = float(V.abs().max() / x.abs().max())
    return rel(direct), rel(wino), mx(direct), mx(wino), vmax


if __name__ == "__main__":
    print("# Winograd F(2x2, 3x3) for the VAE's 3 x 3 x 3 layers: go / no-go (tools/winograd_probe.py, round 5)\n")
    print("## 1. Numerics (measured on the CPU, fp64 direct convolution = reference; 3 x 32 x 32 pixels)\n")
    print("| layer | direct fp16x3: rel. L2 (max / max|ref|) | Winograd F(2x2,3x3) fp16x3: rel. L2 (max / max|ref|) | ratio | largest transformed input / largest input |")
    print("|---|---|---|---|---|")
    for C in (96, 192):
        d, wv, dm, wm_, vm = layer(C)
        print(f"| {C} -> {C} | {d:.2e} ({dm:.2e}) | {wv:.2e} ({wm_:.2e}) | {wv / d:.1f} x | {vm:.2f} |")
    print("""
## 2. Cost model on the shipped kernel's own numbers (`k_conv_w4<0, 3>`, DESIGN section 4 / 4c)

| quantity | direct (shipped) | Winograd F(2x2, 3x3), output-stationary | factor |
|---|---|---|---|
| MFMAs per output pixel and 96 output channels (fp16x3, 16-channel K slices) | 27 taps x 3 terms x (Cin / 16) x 3 N-blocks / 32 px | 16 positions x 3 temporal taps x 3 terms x (Cin / 16) x 3 / (32 tiles x 4 px) | **1 / 2.25** |
| accumulator registers per output pixel block (32 px x 96 ch) | 3 x 16 = 48 | 16 positions x 3 x 16 / 4 = 192 | **4 x** |
| => wave tile that fits 256 accumulators (one wave per SIMD, AGPRs) | 4 pixel blocks x 3 N-blocks (128 px x 96 ch; 192 regs) | 16 positions x 1 block x 1 N-block (32 tiles = 128 px x **32 ch**; 256 regs) | 1/3 of the channels |
| MFMAs per operand fragment read (A from LDS + B weights) | 12 MFMAs per 4 A + 3 B fragments = **1.71** | 1 MFMA per 1 A + 1 B fragment = **0.5** | **3.4 x more fragment traffic per MFMA** |
| LDS A-fragment reads per MFMA | 0.33 (LDS array ~20 % busy, profiles/r2 PMC) | 1.0 (-> ~60 % busy at the same MFMA rate, before bank conflicts) | 3 x |
| weight fragments (global / L2 -> registers, the loop's measured stall source) per MFMA | 0.25 | 1.0, and 16 / 9 more weight bytes per layer | 4 x per MFMA, **1.8 x per output pixel** |
| operand bytes staged per output pixel (patch in LDS: 1.2 px per output px today) | 1.2 x 32 B per 16-ch slice | transformed tile: 16 values per 4 px = **4 x**, so 4.8 x 32 B -- or the transform runs in the loop: 32 fp32 adds + 16 x (split: 2 cvt + 1 sub) VALU per 4 x 4 tile and channel on the lone wave that also issues the MFMAs (it has ~5 issue slots per 32-cycle MFMA and uses 4.1 today, profiles/r4_a_issue_lab.md) | 4 x LDS bytes or > 1 VALU-bound |
| pre-transformed operand in HBM instead (two-pass form) | - | 4 x the [hi, lo] operand: 81 x 480 x 832 x 96 x 4 x 4 B = 50 GB written + read per 96-wide layer = ~20 ms at 5 TB/s, against the layer's whole 18.5 ms today | no |
| error vs fp32 goldens (part 1) | 1 x | see the ratio column | worse, but inside the 40 dB budget |

## 3. Verdict: **no-go** for round 6

The 2.25 x fewer MFMAs are real, and the numerics are acceptable -- but an output-stationary Winograd tile needs 16 accumulator sets for
every 4 output pixels, and `k_conv_w4` already spends the whole register file (256 AGPR-class accumulators + 243 VGPRs, 0 spills) on a 128
px x 96 ch wave tile.  Keeping 256 accumulators means a 32-channel wave tile: every MFMA then needs its own A and B fragment (0.5 MFMAs per
fragment instead of 1.71), i.e. 3.4 x the LDS / L2 -> register traffic per MFMA and 1.5 x per output pixel, on a loop whose measured
remaining loss IS operand delivery (weight loads queued behind LDS-DMA pieces: 95.5 % of the pipe after round 4).  At 2.25 x fewer MFMAs the
loop would have to sustain 3.4 x / 2.25 = 1.5 x today's operand bandwidth per unit time just to break even, with the LDS array going from
~20 % to ~60-70 % busy and the in-order VMEM queue carrying 4 x the weight fragments per MFMA.  The non-fused alternative (transformed
operand through HBM) costs more time than the layer takes today.  F(4x4, 3x3) (4 x fewer MFMAs) needs 36 accumulator sets per 16 pixels
(2.25 x, the same wall) and fractions 1/4 ... 1/24 in its transforms whose error the fp16 split does not absorb.  What WOULD pay on this
layer family is the opposite direction -- more reuse per fragment, not fewer MFMAs -- and that is bounded by the register file as well.
Closed.""")
