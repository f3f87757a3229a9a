"""Oracle: Wan2.1 image-to-video DiT forward in plain torch (CPU, fp32 by default).  TEST INFRASTRUCTURE ONLY.

Restates /root/reference/wan_for_worldforge/wan/modules/model.py (the in-tree statement of diffusers'
WanTransformer3DModel): every function cites the lines it follows.  Weights are a flat dict keyed like the twin's
state_dict ("blocks.3.self_attn.q.weight", "head.head.bias", ...).  Pinned against the imported twin in
tests/test_oracle_dit.py (goldens: tests/golden/g7_dit.npz).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Tuple

import torch
import torch.nn.functional as F


@dataclass
class DiTConfig:
    dim: int = 5120
    ffn_dim: int = 13824
    num_heads: int = 40
    num_layers: int = 40
    in_dim: int = 36
    out_dim: int = 16
    freq_dim: int = 256
    text_dim: int = 4096
    text_len: int = 512
    img_dim: int = 1280
    patch: Tuple[int, int, int] = (1, 2, 2)
    eps: float = 1e-6


def sinusoidal_embedding_1d(dim: int, position: torch.Tensor) -> torch.Tensor:
    """model.py:18-28 (fp64)."""
    half = dim // 2
    position = position.to(torch.float64)
    sinusoid = torch.outer(position, torch.pow(10000, -torch.arange(half).to(position).div(half)))
    return torch.cat([torch.cos(sinusoid), torch.sin(sinusoid)], dim=1)


def rope_tables(head_dim: int, f: int, h: int, w: int, theta: float = 10000.0):
    """model.py:32-39 + 478-485 + 57-62: per-token rotation angles [f*h*w, head_dim/2] (fp64).
    Pair j < d_f uses the frame index, next d_h pairs the row, last d_w the column; d_h = d_w = (head_dim//2)//3."""
    c = head_dim // 2
    d_h = d_w = c // 3
    d_f = c - 2 * (c // 3)

    def axis(n, npairs):
        dim = 2 * npairs
        inv = 1.0 / torch.pow(theta, torch.arange(0, dim, 2).to(torch.float64).div(dim))
        return torch.outer(torch.arange(n).to(torch.float64), inv)  # [n, npairs]

    af, ah, aw = axis(f, d_f), axis(h, d_h), axis(w, d_w)
    ang = torch.cat([af.view(f, 1, 1, -1).expand(f, h, w, -1), ah.view(1, h, 1, -1).expand(f, h, w, -1),
                     aw.view(1, 1, w, -1).expand(f, h, w, -1)], dim=-1)
    return ang.reshape(f * h * w, c)


def rope_apply(x: torch.Tensor, ang: torch.Tensor) -> torch.Tensor:
    """model.py:43-70: x [L, heads, head_dim]; pairs (2j, 2j+1) rotated by ang[:, j] in fp64 -> fp32."""
    L, n, d = x.shape
    xc = x.to(torch.float64).reshape(L, n, d // 2, 2)
    cos, sin = torch.cos(ang).unsqueeze(1), torch.sin(ang).unsqueeze(1)
    re = xc[..., 0] * cos - xc[..., 1] * sin
    im = xc[..., 0] * sin + xc[..., 1] * cos
    return torch.stack([re, im], dim=-1).reshape(L, n, d).float()


def rms_norm(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    """model.py:73-89 (over the full channel dim)."""
    xf = x.float()
    return (xf * torch.rsqrt(xf.pow(2).mean(dim=-1, keepdim=True) + eps)).type_as(x) * weight


def layer_norm(x: torch.Tensor, eps: float, weight=None, bias=None) -> torch.Tensor:
    """model.py:92-102."""
    return F.layer_norm(x.float(), (x.shape[-1],), weight, bias, eps).type_as(x)


def attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """attention.py:24-130 semantics (softmax(q k^T / sqrt(d)) v, no mask): q [Lq,n,d], k/v [Lk,n,d] -> [Lq,n,d]."""
    o = F.scaled_dot_product_attention(q.transpose(0, 1), k.transpose(0, 1), v.transpose(0, 1))
    return o.transpose(0, 1)


def _lin(x, W, prefix):
    return F.linear(x, W[prefix + ".weight"], W.get(prefix + ".bias"))


def self_attention(x, W, p, cfg: DiTConfig, ang):
    """model.py:130-159."""
    L = x.shape[0]
    n, d = cfg.num_heads, cfg.dim // cfg.num_heads
    q = rms_norm(_lin(x, W, p + ".q"), W[p + ".norm_q.weight"], cfg.eps).view(L, n, d)
    k = rms_norm(_lin(x, W, p + ".k"), W[p + ".norm_k.weight"], cfg.eps).view(L, n, d)
    v = _lin(x, W, p + ".v").view(L, n, d)
    o = attention(rope_apply(q, ang), rope_apply(k, ang), v)
    return _lin(o.reshape(L, cfg.dim), W, p + ".o")


def cross_attention_i2v(x, context, W, p, cfg: DiTConfig):
    """model.py:202-229: image tokens are the first (ctx_len - 512) rows."""
    L = x.shape[0]
    n, d = cfg.num_heads, cfg.dim // cfg.num_heads
    n_img = context.shape[0] - cfg.text_len
    c_img, c_txt = context[:n_img], context[n_img:]
    q = rms_norm(_lin(x, W, p + ".q"), W[p + ".norm_q.weight"], cfg.eps).view(L, n, d)
    k = rms_norm(_lin(c_txt, W, p + ".k"), W[p + ".norm_k.weight"], cfg.eps).view(-1, n, d)
    v = _lin(c_txt, W, p + ".v").view(-1, n, d)
    k_img = rms_norm(_lin(c_img, W, p + ".k_img"), W[p + ".norm_k_img.weight"], cfg.eps).view(-1, n, d)
    v_img = _lin(c_img, W, p + ".v_img").view(-1, n, d)
    o = attention(q, k, v).reshape(L, cfg.dim) + attention(q, k_img, v_img).reshape(L, cfg.dim)
    return _lin(o, W, p + ".o")


def block(x, e0, context, W, i, cfg: DiTConfig, ang):
    """model.py:278-317.  x [L,dim] fp32, e0 [6,dim] fp32."""
    p = f"blocks.{i}"
    e = (W[p + ".modulation"][0] + e0).chunk(6, dim=0)
    y = self_attention(layer_norm(x, cfg.eps).float() * (1 + e[1]) + e[0], W, p + ".self_attn", cfg, ang)
    x = x + y * e[2]
    x = x + cross_attention_i2v(layer_norm(x, cfg.eps, W[p + ".norm3.weight"], W[p + ".norm3.bias"]), context, W,
                                p + ".cross_attn", cfg)
    h = layer_norm(x, cfg.eps).float() * (1 + e[4]) + e[3]
    y = _lin(F.gelu(_lin(h, W, p + ".ffn.0"), approximate="tanh"), W, p + ".ffn.2")
    return x + y * e[5]


def embed_condition(t: torch.Tensor, context: torch.Tensor, clip_fea: torch.Tensor, W, cfg: DiTConfig):
    """model.py:546-563: time embedding e [dim], its 6-way projection e0 [6,dim], context [257+512, dim]."""
    e = _lin(F.silu(_lin(sinusoidal_embedding_1d(cfg.freq_dim, t.reshape(1)).float(), W, "time_embedding.0")), W,
             "time_embedding.2")[0]
    e0 = _lin(F.silu(e), W, "time_projection.1").view(6, cfg.dim)
    ctx = torch.cat([context, context.new_zeros(cfg.text_len - context.shape[0], context.shape[1])])
    ctx = _lin(F.gelu(_lin(ctx, W, "text_embedding.0"), approximate="tanh"), W, "text_embedding.2")
    ci = F.layer_norm(clip_fea, (cfg.img_dim,), W["img_emb.proj.0.weight"], W["img_emb.proj.0.bias"])
    ci = _lin(F.gelu(_lin(ci, W, "img_emb.proj.1")), W, "img_emb.proj.3")
    ci = F.layer_norm(ci, (cfg.dim,), W["img_emb.proj.4.weight"], W["img_emb.proj.4.bias"])
    return e, e0, torch.cat([ci, ctx], dim=0)


def patchify(x: torch.Tensor, W, cfg: DiTConfig):
    """model.py:534-537: Conv3d k = s = patch as a linear map on flattened patches -> tokens [L, dim], grid."""
    pt, ph, pw = cfg.patch
    y = F.conv3d(x.unsqueeze(0), W["patch_embedding.weight"], W["patch_embedding.bias"], stride=cfg.patch)
    f, h, w = y.shape[2:]
    return y.flatten(2).transpose(1, 2)[0], (f, h, w)


def head_unpatchify(x, e, W, cfg: DiTConfig, grid):
    """model.py:337-347 + 584-607."""
    em = (W["head.modulation"][0] + e.unsqueeze(0)).chunk(2, dim=0)
    y = _lin(layer_norm(x, cfg.eps) * (1 + em[1]) + em[0], W, "head.head")
    f, h, w = grid
    u = y.view(f, h, w, *cfg.patch, cfg.out_dim)
    u = torch.einsum("fhwpqrc->cfphqwr", u)
    return u.reshape(cfg.out_dim, f * cfg.patch[0], h * cfg.patch[1], w * cfg.patch[2])


def forward(W: Dict[str, torch.Tensor], cfg: DiTConfig, x: torch.Tensor, t: torch.Tensor, context: torch.Tensor,
            clip_fea: torch.Tensor) -> torch.Tensor:
    """model.py:493-582.  x [in_dim(36), T, h, w] (latents ++ condition), t scalar tensor, context [<=512, text_dim],
    clip_fea [257, 1280] -> velocity [16, T, h, w] fp32."""
    tok, grid = patchify(x.float(), W, cfg)
    e, e0, ctx = embed_condition(t, context.float(), clip_fea.float(), W, cfg)
    ang = rope_tables(cfg.dim // cfg.num_heads, *grid)
    for i in range(cfg.num_layers):
        tok = block(tok, e0, ctx, W, i, cfg, ang)
    return head_unpatchify(tok, e, W, cfg, grid).float()


def block_rows(x, rows, e0, context, W, i, cfg: DiTConfig, ang, head_chunk: int = 8):
    """block() (model.py:278-317) for the query rows `rows` only.  Self-attention is global over the sequence (model.py:130-159), so the
    keys and values are still computed from EVERY token; everything after it is row-local.  Used to check one real-width layer at the
    full token count of BASELINE configs[1] / [2], where block() itself would need ~1e14 flop on the CPU.  Pinned against block() in
    tests/test_oracle_dit.py."""
    p = f"blocks.{i}"
    L, R = x.shape[0], len(rows)
    n, d = cfg.num_heads, cfg.dim // cfg.num_heads
    e = (W[p + ".modulation"][0] + e0).chunk(6, dim=0)
    h = layer_norm(x, cfg.eps).float() * (1 + e[1]) + e[0]
    sp = p + ".self_attn"
    q = rms_norm(_lin(h[rows], W, sp + ".q"), W[sp + ".norm_q.weight"], cfg.eps).view(R, n, d)
    k = rms_norm(_lin(h, W, sp + ".k"), W[sp + ".norm_k.weight"], cfg.eps).view(L, n, d)
    v = _lin(h, W, sp + ".v").view(L, n, d)
    o = torch.empty((R, n, d), dtype=torch.float32)
    for h0 in range(0, n, head_chunk):  # bounds the fp64 RoPE temporaries
        hs = slice(h0, min(n, h0 + head_chunk))
        o[:, hs] = attention(rope_apply(q[:, hs], ang[rows]), rope_apply(k[:, hs], ang), v[:, hs])
    xr = x[rows] + _lin(o.reshape(R, cfg.dim), W, sp + ".o") * e[2]
    xr = xr + cross_attention_i2v(layer_norm(xr, cfg.eps, W[p + ".norm3.weight"], W[p + ".norm3.bias"]), context, W, p + ".cross_attn", cfg)
    hh = layer_norm(xr, cfg.eps).float() * (1 + e[4]) + e[3]
    y = _lin(F.gelu(_lin(hh, W, p + ".ffn.0"), approximate="tanh"), W, p + ".ffn.2")
    return xr + y * e[5]


def forward_rows(W: Dict[str, torch.Tensor], cfg: DiTConfig, x: torch.Tensor, t: torch.Tensor, context: torch.Tensor,
                 clip_fea: torch.Tensor, rows) -> torch.Tensor:
    """forward() of a ONE-layer model, evaluated at the tokens `rows` only -> the head's output rows [len(rows), prod(patch) * out_dim]
    in (p_t, p_h, p_w, c) order (what model.py:584-607 un-patchifies): token (f, h, w) covers velocity[c, f + p_t, 2h + p_h, 2w + p_w]."""
    assert cfg.num_layers == 1, "rows of a deeper model depend on every token of the previous layer"
    rows = torch.as_tensor(rows, dtype=torch.long)
    tok, grid = patchify(x.float(), W, cfg)
    e, e0, ctx = embed_condition(t, context.float(), clip_fea.float(), W, cfg)
    ang = rope_tables(cfg.dim // cfg.num_heads, *grid)
    xr = block_rows(tok, rows, e0, ctx, W, 0, cfg, ang)
    em = (W["head.modulation"][0] + e.unsqueeze(0)).chunk(2, dim=0)
    return _lin(layer_norm(xr, cfg.eps) * (1 + em[1]) + em[0], W, "head.head").float()


def token_patches(v: torch.Tensor, cfg: DiTConfig, rows) -> torch.Tensor:
    """The inverse of model.py:584-607 at selected tokens: velocity [out_dim, T, H, W] -> [len(rows), prod(patch) * out_dim] in the
    head's (p_t, p_h, p_w, c) order, for comparing a full forward with forward_rows."""
    pt, ph, pw = cfg.patch
    C, T, Hh, Ww = v.shape
    f, h, w = T // pt, Hh // ph, Ww // pw
    u = v.view(C, f, pt, h, ph, w, pw).permute(1, 3, 5, 2, 4, 6, 0).reshape(f * h * w, pt * ph * pw * C)
    return u[torch.as_tensor(rows, dtype=torch.long)]


def random_weights(cfg: DiTConfig, seed: int = 0, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Synthetic weights of the right shapes: N(0, 0.02^2)-ish linears scaled by fan-in, random head."""
    g = torch.Generator().manual_seed(seed)
    W: Dict[str, torch.Tensor] = {}

    def lin(name, o, i, std=None):
        std = std if std is not None else (1.0 / math.sqrt(i))
        W[name + ".weight"] = (torch.randn(o, i, generator=g) * std).to(dtype)
        W[name + ".bias"] = (torch.randn(o, generator=g) * 0.02).to(dtype)

    d, f = cfg.dim, cfg.ffn_dim
    pt, ph, pw = cfg.patch
    W["patch_embedding.weight"] = (torch.randn(d, cfg.in_dim, pt, ph, pw, generator=g) / math.sqrt(cfg.in_dim * pt * ph * pw)).to(dtype)
    W["patch_embedding.bias"] = (torch.randn(d, generator=g) * 0.02).to(dtype)
    lin("text_embedding.0", d, cfg.text_dim)
    lin("text_embedding.2", d, d)
    lin("time_embedding.0", d, cfg.freq_dim)
    lin("time_embedding.2", d, d)
    lin("time_projection.1", 6 * d, d)
    W["img_emb.proj.0.weight"] = torch.ones(cfg.img_dim, dtype=dtype) + 0.05 * torch.randn(cfg.img_dim, generator=g).to(dtype)
    W["img_emb.proj.0.bias"] = (0.02 * torch.randn(cfg.img_dim, generator=g)).to(dtype)
    lin("img_emb.proj.1", cfg.img_dim, cfg.img_dim)
    lin("img_emb.proj.3", d, cfg.img_dim)
    W["img_emb.proj.4.weight"] = torch.ones(d, dtype=dtype) + 0.05 * torch.randn(d, generator=g).to(dtype)
    W["img_emb.proj.4.bias"] = (0.02 * torch.randn(d, generator=g)).to(dtype)
    for i in range(cfg.num_layers):
        p = f"blocks.{i}"
        for a in ("self_attn", "cross_attn"):
            for nm in ("q", "k", "v", "o"):
                lin(f"{p}.{a}.{nm}", d, d)
            W[f"{p}.{a}.norm_q.weight"] = (1 + 0.05 * torch.randn(d, generator=g)).to(dtype)
            W[f"{p}.{a}.norm_k.weight"] = (1 + 0.05 * torch.randn(d, generator=g)).to(dtype)
        lin(f"{p}.cross_attn.k_img", d, d)
        lin(f"{p}.cross_attn.v_img", d, d)
        W[f"{p}.cross_attn.norm_k_img.weight"] = (1 + 0.05 * torch.randn(d, generator=g)).to(dtype)
        W[f"{p}.norm3.weight"] = (1 + 0.05 * torch.randn(d, generator=g)).to(dtype)
        W[f"{p}.norm3.bias"] = (0.02 * torch.randn(d, generator=g)).to(dtype)
        lin(f"{p}.ffn.0", f, d)
        lin(f"{p}.ffn.2", d, f)
        W[f"{p}.modulation"] = (torch.randn(1, 6, d, generator=g) / math.sqrt(d)).to(dtype)
    lin("head.head", cfg.out_dim * pt * ph * pw, d)
    W["head.modulation"] = (torch.randn(1, 2, d, generator=g) / math.sqrt(d)).to(dtype)
    return W
