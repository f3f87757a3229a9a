"""Oracle: LongCat-Video DiT forward in plain torch (CPU, fp32).  TEST INFRASTRUCTURE ONLY.

Restates /root/reference/longcat_for_worldforge/longcat_video/modules/{longcat_video_dit.py, attention.py, blocks.py,
rope_3d.py} (abbreviated LCD / LCA / LCB / LCR below) for the call the guided image-to-video sampler makes
(pipeline_longcat_video.py:867-873: one sample, per-frame timesteps with frame 0 clean, num_cond_latents = 1, no KV cache, no
block-sparse attention).  Weights are a flat dict keyed like the reference state_dict ("blocks.3.attn.qkv.weight", ...).
Pinned against the imported, unmodified reference module in tests/test_oracle_longcat.py (goldens: tests/golden/g11_longcat_dit.npz,
written by tools/make_goldens.py longcat; the reference's flash-attn calls are served there by a plain-softmax stand-in for that
third-party package, tools/refshim/flash_attn).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Tuple

import numpy as np
import torch
import torch.nn.functional as F


@dataclass
class LongCatConfig:
    """LCD:138-158 defaults = the released 13.6B model."""
    hidden_size: int = 4096
    depth: int = 48
    num_heads: int = 32
    in_channels: int = 16
    out_channels: int = 16
    caption_channels: int = 4096
    mlp_ratio: int = 4
    adaln_tembed_dim: int = 512
    frequency_embedding_size: int = 256
    patch_size: Tuple[int, int, int] = (1, 2, 2)
    text_tokens_zero_pad: bool = False

    @property
    def ffn_hidden(self) -> int:
        """LCB:17-29: SwiGLU width = 2/3 of mlp_ratio * hidden, rounded up to a multiple of 256."""
        h = int(2 * int(self.hidden_size * self.mlp_ratio) / 3)
        return 256 * ((h + 255) // 256)


def timestep_embedding(t: torch.Tensor, dim: int, max_period: float = 10000.0) -> torch.Tensor:
    """LCB:181-199: [cos | sin] of t * exp(-ln(max_period) * i / half), fp32."""
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def rope_angles(head_dim: int, f: int, h: int, w: int, base: float = 10000.0) -> torch.Tensor:
    """LCR:68-99: per-token angles [f*h*w, head_dim], every frequency repeated for the two members of its pair; the first
    head_dim - 4*(head_dim//6) entries follow the frame index, then 2*(head_dim//6) the row, then the column (fp32)."""
    d_hw = 2 * (head_dim // 6)
    d_t = head_dim - 2 * d_hw

    def axis(n, dim):
        freqs = 1.0 / (base ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim))
        grid = torch.from_numpy(np.linspace(0, n, n, endpoint=False, dtype=np.float32)).float()
        return torch.outer(grid, freqs).repeat_interleave(2, dim=-1)  # [n, dim]

    at, ah, aw = axis(f, d_t), axis(h, d_hw), axis(w, d_hw)
    ang = torch.cat([at.view(f, 1, 1, -1).expand(f, h, w, -1), ah.view(1, h, 1, -1).expand(f, h, w, -1),
                     aw.view(1, 1, w, -1).expand(f, h, w, -1)], dim=-1)
    return ang.reshape(f * h * w, head_dim)


def rope_apply(x: torch.Tensor, ang: torch.Tensor) -> torch.Tensor:
    """LCR:32-36 + 116-120: x [heads, L, D]; x*cos + rotate_half(x)*sin with interleaved pairs (x0, x1) -> (-x1, x0), fp32."""
    xf = x.float()
    cos, sin = ang.cos()[None], ang.sin()[None]
    x2 = xf.reshape(*xf.shape[:-1], -1, 2)
    rot = torch.stack((-x2[..., 1], x2[..., 0]), dim=-1).reshape(xf.shape)
    return (xf * cos + rot * sin).type_as(x)


def rms_norm_head(x: torch.Tensor, weight: torch.Tensor, eps: float = 1e-6) -> torch.Tensor:
    """LCB:40-52: RMS over the last (head) dimension in fp32, cast back, times weight."""
    xf = x.float()
    return (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps)).type_as(x) * weight


def layer_norm(x: torch.Tensor, weight=None, bias=None, eps: float = 1e-6) -> torch.Tensor:
    """LCB:55-68."""
    return F.layer_norm(x.float(), (x.shape[-1],), None if weight is None else weight.float(),
                        None if bias is None else bias.float(), eps).to(x.dtype)


def modulate(x: torch.Tensor, shift: torch.Tensor, scale: torch.Tensor, tokens_per_frame: int) -> torch.Tensor:
    """LCB:133-141 with the per-frame [T, C] parameters of LCD:85-91 broadcast over the frame's tokens."""
    L, C = x.shape
    xn = layer_norm(x.float()).view(-1, tokens_per_frame, C)
    return (xn * (scale[:, None, :] + 1) + shift[:, None, :]).view(L, C).to(x.dtype)


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: float) -> torch.Tensor:
    """softmax(q k^T * scale) v per head (what LCA:70-90 asks of flash-attn): q [H, Lq, D], k/v [H, Lk, D] -> [Lq, H*D]."""
    s = torch.einsum("hqd,hkd->hqk", q.float(), k.float()) * scale
    o = torch.einsum("hqk,hkd->hqd", torch.softmax(s, dim=-1), v.float())
    return o.permute(1, 0, 2).reshape(q.shape[1], -1).to(q.dtype)


def self_attention(W, p: str, x: torch.Tensor, cfg: LongCatConfig, ang: torch.Tensor, n_cond_tokens: int, grid=None, bsa=None,
                   picked=None) -> torch.Tensor:
    """LCA:105-145: fused qkv -> per-head RMS norm of q, k -> 3D RoPE -> condition tokens attend to condition tokens only, noise
    tokens to everything -> proj."""
    L, C = x.shape
    H = cfg.num_heads
    D = C // H
    qkv = F.linear(x, W[p + "qkv.weight"], W[p + "qkv.bias"]).view(L, 3, H, D).permute(1, 2, 0, 3)  # [3, H, L, D]
    q, k, v = qkv[0], qkv[1], qkv[2]
    q, k = rms_norm_head(q, W[p + "q_norm.weight"]), rms_norm_head(k, W[p + "k_norm.weight"])
    q, k = rope_apply(q, ang), rope_apply(k, ang)
    scale = D ** -0.5
    if bsa is not None and grid[0] > 1:
        # LCA:57-66: block-sparse attention (oracle/bsa.py); the gating runs in `gate_dtype` (bf16 in the reference's bf16 model)
        from . import bsa as obsa
        T, gh, gw = grid
        tpf = gh * gw
        gd = bsa.get("gate_dtype", torch.float32)

        def part(qq, kk, vv):
            sq, sk = (qq.shape[1] // tpf, gh, gw), (kk.shape[1] // tpf, gh, gw)
            if picked is None:
                out, idx = obsa.flash_attn_bsa_3d(qq, kk, vv, sq, sk, bsa["sparsity"], bsa["chunk_3d_shape_q"], bsa["chunk_3d_shape_k"],
                                                  return_indices=True, gate_dtype=gd)
            else:
                out = obsa.flash_attn_bsa_3d(qq, kk, vv, sq, sk, bsa["sparsity"], bsa["chunk_3d_shape_q"], bsa["chunk_3d_shape_k"],
                                             block_indices=picked.pop(0))
            return out.permute(1, 0, 2).reshape(qq.shape[1], -1)

        nc = n_cond_tokens
        o = torch.cat([part(q[:, :nc], k[:, :nc], v[:, :nc]), part(q[:, nc:], k, v)], dim=0) if nc > 0 else part(q, k, v)
    elif n_cond_tokens > 0:
        nc = n_cond_tokens
        o = torch.cat([attention(q[:, :nc], k[:, :nc], v[:, :nc], scale), attention(q[:, nc:], k, v, scale)], dim=0)
    else:
        o = attention(q, k, v, scale)
    return F.linear(o, W[p + "proj.weight"], W[p + "proj.bias"])


def cross_attention(W, p: str, x: torch.Tensor, y: torch.Tensor, cfg: LongCatConfig, n_cond_tokens: int) -> torch.Tensor:
    """LCA:218-276: q from the noise tokens only, k / v from the valid caption tokens, per-head RMS norm of q and k, default
    softmax scale 1/sqrt(D); condition tokens receive zeros."""
    L, C = x.shape
    H = cfg.num_heads
    D = C // H
    xn = x[n_cond_tokens:]
    q = F.linear(xn, W[p + "q_linear.weight"], W[p + "q_linear.bias"]).view(-1, H, D)
    kv = F.linear(y, W[p + "kv_linear.weight"], W[p + "kv_linear.bias"]).view(-1, 2, H, D)
    k, v = kv[:, 0], kv[:, 1]
    q, k = rms_norm_head(q, W[p + "q_norm.weight"]), rms_norm_head(k, W[p + "k_norm.weight"])
    o = attention(q.transpose(0, 1), k.transpose(0, 1), v.transpose(0, 1), D ** -0.5)
    o = F.linear(o, W[p + "proj.weight"], W[p + "proj.bias"])
    return torch.cat([torch.zeros((n_cond_tokens, C), dtype=o.dtype), o], dim=0)


def swiglu(W, p: str, x: torch.Tensor) -> torch.Tensor:
    """LCB:36-37."""
    return F.linear(F.silu(F.linear(x, W[p + "w1.weight"])) * F.linear(x, W[p + "w3.weight"]), W[p + "w2.weight"])


def block(W, i: int, x: torch.Tensor, y: torch.Tensor, t: torch.Tensor, cfg: LongCatConfig, ang: torch.Tensor,
          tokens_per_frame: int, n_cond_tokens: int, grid=None, bsa=None, picked=None) -> torch.Tensor:
    """LCD:68-121.  x [L, C]; y [n_valid, C]; t [T, C_t] fp32."""
    p = f"blocks.{i}."
    C = cfg.hidden_size
    mod = F.linear(F.silu(t.float()), W[p + "adaLN_modulation.1.weight"].float(), W[p + "adaLN_modulation.1.bias"].float())
    shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = mod.chunk(6, dim=-1)  # each [T, C]

    def gated(x, gate, xs):
        return (x.float().view(-1, tokens_per_frame, C) + gate[:, None, :] * xs.float().view(-1, tokens_per_frame, C)) \
            .view(-1, C).to(x.dtype)

    xs = self_attention(W, p + "attn.", modulate(x, shift_msa, scale_msa, tokens_per_frame), cfg, ang, n_cond_tokens, grid, bsa, picked)
    x = gated(x, gate_msa, xs)
    xn = layer_norm(x, W[p + "pre_crs_attn_norm.weight"], W[p + "pre_crs_attn_norm.bias"])
    x = x + cross_attention(W, p + "cross_attn.", xn, y, cfg, n_cond_tokens)
    xs = swiglu(W, p + "ffn.", modulate(x, shift_mlp, scale_mlp, tokens_per_frame))
    return gated(x, gate_mlp, xs)


def forward(W: Dict[str, torch.Tensor], cfg: LongCatConfig, latents: torch.Tensor, timesteps: torch.Tensor, caption: torch.Tensor,
            caption_mask: torch.Tensor = None, num_cond_latents: int = 0, bsa: dict = None, bsa_indices=None) -> torch.Tensor:
    """LCD:279-366 for one sample.  latents [C_in, T, H, W]; timesteps [T] (one per latent frame, LCD:299-301); caption
    [n_tokens, caption_channels]; caption_mask [n_tokens] (0 = padding) or None -> velocity [C_out, T, H, W] fp32."""
    Cin, T, Hh, Ww = latents.shape
    pt, ph, pw = cfg.patch_size
    assert pt == 1
    nh, nw = Hh // ph, Ww // pw
    C = cfg.hidden_size
    # LCB:112 Conv3d with kernel = stride = patch -> tokens in (t, h, w) order
    x = F.conv3d(latents[None].float(), W["x_embedder.proj.weight"].float(), W["x_embedder.proj.bias"].float(), stride=cfg.patch_size)
    x = x.flatten(2).transpose(1, 2)[0]  # [L, C]
    # LCB:201-206 + LCD:310-311
    tf = timestep_embedding(timesteps.float().flatten(), cfg.frequency_embedding_size)
    t = F.linear(F.silu(F.linear(tf, W["t_embedder.mlp.0.weight"].float(), W["t_embedder.mlp.0.bias"].float())),
                 W["t_embedder.mlp.2.weight"].float(), W["t_embedder.mlp.2.bias"].float())  # [T, C_t]
    # LCB:225-228 + LCD:315-325
    y = F.linear(F.gelu(F.linear(caption, W["y_embedder.y_proj.0.weight"], W["y_embedder.y_proj.0.bias"]), approximate="tanh"),
                 W["y_embedder.y_proj.2.weight"], W["y_embedder.y_proj.2.bias"])
    if caption_mask is not None:
        if cfg.text_tokens_zero_pad:
            y = y * caption_mask[:, None].to(y.dtype)
        else:
            y = y[caption_mask != 0]
    ang = rope_angles(C // cfg.num_heads, T, nh, nw)
    tpf = nh * nw
    for i in range(cfg.depth):
        # bsa: dict(sparsity, chunk_3d_shape_q, chunk_3d_shape_k[, gate_dtype]) enables the block-sparse self-attention of the refine
        # pass (LCD:270-272); bsa_indices: per block a list of block-index tensors to use instead of the oracle's own top-k
        x = block(W, i, x, y, t, cfg, ang, tpf, num_cond_latents * tpf, (T, nh, nw), bsa,
                  list(bsa_indices[i]) if bsa_indices is not None else None)
    # LCB:159-168
    mod = F.linear(F.silu(t), W["final_layer.adaLN_modulation.1.weight"].float(), W["final_layer.adaLN_modulation.1.bias"].float())
    shift, scale = mod.chunk(2, dim=-1)
    x = modulate(x, shift, scale, tpf)
    x = F.linear(x.float(), W["final_layer.linear.weight"].float(), W["final_layer.linear.bias"].float())
    # LCD:371-392: [L, (ph pw C_out)] -> [C_out, T, H, W]
    x = x.view(T, nh, nw, ph, pw, cfg.out_channels).permute(5, 0, 1, 3, 2, 4).reshape(cfg.out_channels, T, nh * ph, nw * pw)
    return x.float()


def forward_rows(W: Dict[str, torch.Tensor], cfg: LongCatConfig, latents: torch.Tensor, timesteps: torch.Tensor, caption: torch.Tensor,
                 caption_mask: torch.Tensor, num_cond_latents: int, rows) -> torch.Tensor:
    """forward() of a ONE-block model at the tokens `rows` only (dense attention) -> the final layer's rows [len(rows), p_h p_w C_out] in
    (p_h, p_w, c) order.  The keys / values of the self-attention (LCA:105-145) still come from every token; condition rows see the
    condition keys only (LCA:123-138) and receive no cross-attention (LCA:262-276).  For the GPU check of one released-width block at the
    37 440-token grid of BASELINE configs[3]; pinned against forward() in tests/test_oracle_longcat.py."""
    assert cfg.depth == 1, "rows of a deeper model depend on every token of the previous block"
    rows = torch.as_tensor(rows, dtype=torch.long)
    Cin, T, Hh, Ww = latents.shape
    pt, ph, pw = cfg.patch_size
    nh, nw = Hh // ph, Ww // pw
    C, H = cfg.hidden_size, cfg.num_heads
    D = C // H
    tpf = nh * nw
    nc = num_cond_latents * tpf
    x = F.conv3d(latents[None].float(), W["x_embedder.proj.weight"].float(), W["x_embedder.proj.bias"].float(), stride=cfg.patch_size)
    x = x.flatten(2).transpose(1, 2)[0]
    L = x.shape[0]
    tf = timestep_embedding(timesteps.float().flatten(), cfg.frequency_embedding_size)
    t = F.linear(F.silu(F.linear(tf, W["t_embedder.mlp.0.weight"].float(), W["t_embedder.mlp.0.bias"].float())),
                 W["t_embedder.mlp.2.weight"].float(), W["t_embedder.mlp.2.bias"].float())
    y = F.linear(F.gelu(F.linear(caption, W["y_embedder.y_proj.0.weight"], W["y_embedder.y_proj.0.bias"]), approximate="tanh"),
                 W["y_embedder.y_proj.2.weight"], W["y_embedder.y_proj.2.bias"])
    if caption_mask is not None:
        y = y * caption_mask[:, None].to(y.dtype) if cfg.text_tokens_zero_pad else y[caption_mask != 0]
    ang = rope_angles(D, T, nh, nw)
    p = "blocks.0."
    mod = F.linear(F.silu(t.float()), W[p + "adaLN_modulation.1.weight"].float(), W[p + "adaLN_modulation.1.bias"].float())
    shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = mod.chunk(6, dim=-1)
    frame = rows // tpf

    def mod_rows(xr, shift, scale):  # modulate() on selected rows
        return (layer_norm(xr.float()) * (scale[frame] + 1) + shift[frame]).to(xr.dtype)

    # self-attention: k, v from every token, q from the sampled rows
    hm = modulate(x, shift_msa, scale_msa, tpf)
    wq, bq = W[p + "attn.qkv.weight"], W[p + "attn.qkv.bias"]
    kv = F.linear(hm, wq[C:], bq[C:]).view(L, 2, H, D).permute(1, 2, 0, 3)
    k, v = rope_apply(rms_norm_head(kv[0], W[p + "attn.k_norm.weight"]), ang), kv[1]
    q = F.linear(hm[rows], wq[:C], bq[:C]).view(-1, H, D).permute(1, 0, 2)
    q = rope_apply(rms_norm_head(q, W[p + "attn.q_norm.weight"]), ang[rows])
    is_c = rows < nc
    o = torch.empty((len(rows), C), dtype=x.dtype)
    if is_c.any():
        o[is_c] = attention(q[:, is_c], k[:, :nc], v[:, :nc], D ** -0.5)
    if (~is_c).any():
        o[~is_c] = attention(q[:, ~is_c], k, v, D ** -0.5)
    xs = F.linear(o, W[p + "attn.proj.weight"], W[p + "attn.proj.bias"])
    xr = (x[rows].float() + gate_msa[frame] * xs.float()).to(x.dtype)
    # cross-attention on the noise rows (condition rows receive zeros)
    xn = layer_norm(xr, W[p + "pre_crs_attn_norm.weight"], W[p + "pre_crs_attn_norm.bias"])
    cp = p + "cross_attn."
    qc = F.linear(xn[~is_c], W[cp + "q_linear.weight"], W[cp + "q_linear.bias"]).view(-1, H, D)
    kvc = F.linear(y, W[cp + "kv_linear.weight"], W[cp + "kv_linear.bias"]).view(-1, 2, H, D)
    qc, kc = rms_norm_head(qc, W[cp + "q_norm.weight"]), rms_norm_head(kvc[:, 0], W[cp + "k_norm.weight"])
    oc = attention(qc.transpose(0, 1), kc.transpose(0, 1), kvc[:, 1].transpose(0, 1), D ** -0.5)
    ca = torch.zeros_like(xr)
    ca[~is_c] = F.linear(oc, W[cp + "proj.weight"], W[cp + "proj.bias"])
    xr = xr + ca
    xs = swiglu(W, p + "ffn.", mod_rows(xr, shift_mlp, scale_mlp))
    xr = (xr.float() + gate_mlp[frame] * xs.float()).to(xr.dtype)
    mod = F.linear(F.silu(t), W["final_layer.adaLN_modulation.1.weight"].float(), W["final_layer.adaLN_modulation.1.bias"].float())
    shift, scale = mod.chunk(2, dim=-1)
    return F.linear(mod_rows(xr, shift, scale).float(), W["final_layer.linear.weight"].float(), W["final_layer.linear.bias"].float()).float()


def token_patches(v: torch.Tensor, cfg: LongCatConfig, rows) -> torch.Tensor:
    """The inverse of LCD:371-392 at selected tokens: velocity [C_out, T, H, W] -> [len(rows), p_h p_w C_out] in (p_h, p_w, c) order."""
    _, ph, pw = cfg.patch_size
    Co, T, Hh, Ww = v.shape
    nh, nw = Hh // ph, Ww // pw
    u = v.view(Co, T, nh, ph, nw, pw).permute(1, 2, 4, 3, 5, 0).reshape(T * nh * nw, ph * pw * Co)
    return u[torch.as_tensor(rows, dtype=torch.long)]


def random_weights(cfg: LongCatConfig, seed: int = 0, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Synthetic state dict with the reference's names and shapes; values are bf16-representable so that the HIP path (bf16
    weights) and this fp32 oracle see identical parameters."""
    g = torch.Generator().manual_seed(seed)
    C, Ct, Hd = cfg.hidden_size, cfg.adaln_tembed_dim, cfg.ffn_hidden
    D = C // cfg.num_heads
    W = {}

    def r(*shape, std=None, base=0.0):
        std = std if std is not None else 1.0 / math.sqrt(shape[-1])
        return (torch.randn(*shape, generator=g) * std + base).to(torch.bfloat16).to(dtype)

    def lin(name, o, i, bias=True, std=None):
        W[name + ".weight"] = r(o, i, std=std)
        if bias:
            W[name + ".bias"] = r(o, std=0.02)

    W["x_embedder.proj.weight"] = r(C, cfg.in_channels, *cfg.patch_size, std=1.0 / math.sqrt(cfg.in_channels * 4))
    W["x_embedder.proj.bias"] = r(C, std=0.02)
    lin("t_embedder.mlp.0", Ct, cfg.frequency_embedding_size)
    lin("t_embedder.mlp.2", Ct, Ct)
    lin("y_embedder.y_proj.0", C, cfg.caption_channels)
    lin("y_embedder.y_proj.2", C, C)
    for i in range(cfg.depth):
        p = f"blocks.{i}."
        lin(p + "adaLN_modulation.1", 6 * C, Ct, std=0.5 / math.sqrt(Ct))
        W[p + "pre_crs_attn_norm.weight"] = r(C, std=0.05, base=1.0)
        W[p + "pre_crs_attn_norm.bias"] = r(C, std=0.02)
        lin(p + "attn.qkv", 3 * C, C)
        lin(p + "attn.proj", C, C)
        lin(p + "cross_attn.q_linear", C, C)
        lin(p + "cross_attn.kv_linear", 2 * C, C)
        lin(p + "cross_attn.proj", C, C)
        for a in ("attn", "cross_attn"):
            W[p + a + ".q_norm.weight"] = r(D, std=0.05, base=1.0)
            W[p + a + ".k_norm.weight"] = r(D, std=0.05, base=1.0)
        lin(p + "ffn.w1", Hd, C, bias=False)
        lin(p + "ffn.w2", C, Hd, bias=False)
        lin(p + "ffn.w3", Hd, C, bias=False)
    lin("final_layer.linear", 4 * cfg.out_channels, C)
    lin("final_layer.adaLN_modulation.1", 2 * C, Ct, std=0.5 / math.sqrt(Ct))
    return W


def fold_lora(W: Dict[str, torch.Tensor], lora: Dict[str, torch.Tensor], multiplier: float = 1.0, network_dim: int = 128,
              network_alpha: float = 64.0) -> Dict[str, torch.Tensor]:
    """Runtime LoRA of LCD:189-247 + lora_utils.py:27-78 as a weight update: for every `<name>.lora_down.weight` in the LoRA state dict
    (name = "lora___lorahyphen___" + module path with "." written "___lorahyphen___"), the wrapped Linear computes
    org(x) + multiplier * alpha_scale * up(down(x)), i.e. W' = W + multiplier * alpha_scale * U @ D, where U is `lora_up.weight`
    or, for n separate up-blocks (lora_utils.py:15-24), the block-diagonal stack of `lora_up.blocks.i.weight` acting on the i-th
    rank-slice of D.  alpha_scale = the stored buffer if present, else alpha / dim."""
    out = dict(W)
    for key in lora:
        if not key.endswith(".lora_down.weight"):
            continue
        name = key[: -len(".lora_down.weight")]
        module = name.replace("lora___lorahyphen___", "").replace("___lorahyphen___", ".")
        down = lora[key].float()
        scale = float(lora[name + ".alpha_scale"]) if name + ".alpha_scale" in lora else (network_alpha or network_dim) / network_dim
        if name + ".lora_up.weight" in lora:
            delta = lora[name + ".lora_up.weight"].float() @ down
        else:
            blocks = sorted((k for k in lora if k.startswith(name + ".lora_up.blocks.")), key=lambda k: int(k.split(".")[-2]))
            r = down.shape[0] // len(blocks)
            delta = torch.cat([lora[k].float() @ down[i * r:(i + 1) * r] for i, k in enumerate(blocks)], dim=0)
        wk = module + ".weight"
        out[wk] = (W[wk].float() + multiplier * scale * delta).to(W[wk].dtype)
    return out
