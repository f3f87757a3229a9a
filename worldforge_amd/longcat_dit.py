"""LongCat-Video DiT (LongCatVideoTransformer3DModel) on hand-written HIP kernels.

Speaks the call protocol of the reference sampler (longcat_video/pipeline_longcat_video.py:867-873):
    dit(hidden_states[B,16,T,H,W], timestep[B,T], encoder_hidden_states[B,1,N,4096], encoder_attention_mask[B,N],
        num_cond_latents=1) -> fp32 [B,16,T,H,W]
and follows longcat_video/modules/longcat_video_dit.py (LCD), attention.py (LCA), blocks.py (LCB), rope_3d.py (LCR); file:line
citations are on the kernels (csrc/longcat_ops.hip, gemm.hip, attention.hip) and below.  Every FLOP of the token path runs in
libwf_hip.so; PyTorch owns the buffers.  The samples of a batch (the CFG pair of pipeline:857-866) are run one after the other:
the reference's varlen / block-diagonal cross-attention (LCA:236-262) never mixes samples.

Numerics: bf16 weights, bf16 residual stream (LCD:104, 120), bf16 GEMM / attention operands with fp32 accumulation, fp32 timestep
embedding and AdaLN parameters (LCD:84-88, 310-311: the fp32 activations are fed to the bf16 MFMA GEMM as a hi + lo bf16 pair, which
keeps 16 mantissa bits), fp32 LayerNorm statistics, fp32 final projection (LCB:162-167).

Runtime LoRA (LCD:189-268) is offered as a weight fold at load (`fold_lora`).  Not covered here (next rows of SURVEY section 8f):
KV-cache continuation (LCA:147-181), block-sparse attention (LCA:57-66), sequence parallelism.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass
from types import SimpleNamespace
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import ops
from ._ffi import WF_BF16, WF_F32, call
from .dit import EPI_BF16, EPI_BF16_GELU, EPI_F32, EPI_F32_ACC, _pad64, attention, attention_exchange, gemm, head_max_norm2


@dataclass
class LongCatConfig:
    """LCD:138-158 defaults = the released model."""
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
    eps: float = 1e-6

    @property
    def ffn_hidden(self) -> int:
        """LCB:17-29."""
        h = int(2 * int(self.hidden_size * self.mlp_ratio) / 3)
        return 256 * ((h + 255) // 256)


def timestep_embedding(ts, dim: int, max_period: float = 10000.0) -> torch.Tensor:
    """LCB:181-199 on the host in fp32: [cos | sin] -> [len(ts), dim]."""
    half = dim // 2
    freqs = np.exp(np.float32(-math.log(max_period)) * np.arange(half, dtype=np.float32) / np.float32(half)).astype(np.float32)
    args = np.asarray(ts, dtype=np.float32)[:, None] * freqs[None]
    emb = np.concatenate([np.cos(args), np.sin(args)], axis=-1).astype(np.float32)
    if dim % 2:
        emb = np.concatenate([emb, np.zeros_like(emb[:, :1])], axis=-1)
    return torch.from_numpy(emb)


def rope_tables(head_dim: int, f: int, h: int, w: int, base: float = 10000.0):
    """LCR:68-99 + 113-115: cos / sin [f*h*w, head_dim/2] fp32, one entry per rotation pair (2p, 2p+1): the first
    (head_dim - 4*(head_dim//6))/2 pairs turn with the frame index, the next head_dim//6 with the row, the last with the column."""
    d_hw = 2 * (head_dim // 6)
    d_t = head_dim - 2 * d_hw

    def axis(n, dim):
        freqs = 1.0 / (base ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim))
        grid = torch.from_numpy(np.linspace(0, n, n, endpoint=False, dtype=np.float32)).float()
        return torch.outer(grid, freqs)  # [n, dim/2]

    at, ah, aw = axis(f, d_t), axis(h, d_hw), axis(w, d_hw)
    ang = torch.cat([at.view(f, 1, 1, -1).expand(f, h, w, -1), ah.view(1, h, 1, -1).expand(f, h, w, -1),
                     aw.view(1, 1, w, -1).expand(f, h, w, -1)], dim=-1).reshape(f * h * w, head_dim // 2)
    return ang.cos().contiguous(), ang.sin().contiguous()


def fold_lora(sd: Dict[str, torch.Tensor], lora_sd: Dict[str, torch.Tensor], multiplier: float = 1.0, network_dim: int = 128,
              network_alpha: float = 64.0) -> Dict[str, torch.Tensor]:
    """The reference applies LoRA at run time (LCD:189-247, lora_utils.py:27-78): every wrapped Linear returns
    org(x) + multiplier * alpha_scale * up(down(x)).  Here the update is folded into the weights once, at load (fp32 accumulate, then
    the usual bf16 storage): W' = W + multiplier * alpha_scale * U @ D, with U block-diagonal over the rank slices of D when the
    up-projection is stored as n separate blocks (fused qkv / kv).  A weight-load step, not part of the sampling path; the forward then
    costs exactly what the base model costs.  Returns a new reference-keyed state dict for load_state_dict."""
    out = dict(sd)
    for key in lora_sd:
        if not key.endswith(".lora_down.weight"):
            continue
        name = key[: -len(".lora_down.weight")]
        module = name.replace("lora___lorahyphen___", "").replace("___lorahyphen___", ".")
        wk = module + ".weight"
        if wk not in sd:
            raise KeyError(f"LoRA entry {name} has no Linear {wk} in the model")
        dev = sd[wk].device
        down = lora_sd[key].to(dev, torch.float32)
        if name + ".alpha_scale" in lora_sd:
            scale = float(lora_sd[name + ".alpha_scale"])
        else:
            scale = (network_alpha or network_dim) / network_dim
        if name + ".lora_up.weight" in lora_sd:
            delta = lora_sd[name + ".lora_up.weight"].to(dev, torch.float32) @ down
        else:
            blocks = sorted((k for k in lora_sd if k.startswith(name + ".lora_up.blocks.")), key=lambda k: int(k.split(".")[-2]))
            if not blocks:
                raise KeyError(f"LoRA entry {name} has no up-projection")
            r = down.shape[0] // len(blocks)
            delta = torch.cat([lora_sd[k].to(dev, torch.float32) @ down[i * r:(i + 1) * r] for i, k in enumerate(blocks)], dim=0)
        if tuple(delta.shape) != tuple(sd[wk].shape):
            raise ValueError(f"LoRA update for {wk} has shape {tuple(delta.shape)}, weight is {tuple(sd[wk].shape)}")
        out[wk] = sd[wk].to(torch.float32) + (multiplier * scale) * delta
    return out


class LongCatVideoTransformer3DModel:
    dtype = torch.bfloat16

    def __init__(self, cfg: LongCatConfig, device="cuda:0", enable_bsa: bool = False, bsa_params: Optional[dict] = None, comm=None):
        assert cfg.hidden_size // cfg.num_heads == 128 and cfg.hidden_size % cfg.num_heads == 0, "attention kernel is built for head_dim 128"
        assert cfg.patch_size == (1, 2, 2)
        self.cfg = cfg
        self.config = SimpleNamespace(in_channels=cfg.in_channels, out_channels=cfg.out_channels, patch_size=cfg.patch_size)
        self.cp_split_hw = None
        self.comm = comm  # parallel.Comm (or a stand-in): sequence parallelism over contiguous token shards, see forward_tokens
        self.device = torch.device(device)
        self.w: Dict[str, torch.Tensor] = {}
        self._ws = {}
        self._rope = {}
        # block-sparse self-attention of the 720p refine pass (LCA:57-66, LCD:270-276); bsa_params as in the reference's config:
        # sparsity, chunk_3d_shape_q, chunk_3d_shape_k (cdf_threshold is not built)
        self._bsa = bool(enable_bsa)
        self.bsa_params = dict(bsa_params) if bsa_params else dict(sparsity=0.875, chunk_3d_shape_q=[4, 4, 8], chunk_3d_shape_k=[4, 4, 8])  # bsa_interface.py:618-621 defaults
        self.last_bsa_indices = None

    def enable_bsa(self):
        """LCD:270-272."""
        self._bsa = True

    def disable_bsa(self):
        """LCD:274-276."""
        self._bsa = False

    # ------------------------------------------------------------------------------------------------------------
    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        """Reference-keyed state dict (any dtype / device) -> device tensors: matrices bf16, vectors fp32; the AdaLN projections of
        all blocks stacked into one matrix, w1 | w3 fused."""
        cfg, dev = self.cfg, self.device
        W = {}
        mat = lambda k: sd[k].to(device=dev, dtype=torch.bfloat16).contiguous()  # noqa: E731
        vec = lambda k: sd[k].to(device=dev, dtype=torch.float32).contiguous()  # noqa: E731
        W["patch.w"] = mat("x_embedder.proj.weight").reshape(cfg.hidden_size, -1).contiguous()
        W["patch.b"] = vec("x_embedder.proj.bias")
        for n in ("t_embedder.mlp.0", "t_embedder.mlp.2", "y_embedder.y_proj.0", "y_embedder.y_proj.2", "final_layer.linear",
                  "final_layer.adaLN_modulation.1"):
            W[n + ".w"], W[n + ".b"] = mat(n + ".weight"), vec(n + ".bias")
        W["ada.w"] = torch.cat([mat(f"blocks.{i}.adaLN_modulation.1.weight") for i in range(cfg.depth)], 0).contiguous()
        W["ada.b"] = torch.cat([vec(f"blocks.{i}.adaLN_modulation.1.bias") for i in range(cfg.depth)], 0).contiguous()
        for i in range(cfg.depth):
            p = f"blocks.{i}."
            W[p + "norm.w"], W[p + "norm.b"] = vec(p + "pre_crs_attn_norm.weight"), vec(p + "pre_crs_attn_norm.bias")
            for n in ("attn.qkv", "attn.proj", "cross_attn.q_linear", "cross_attn.kv_linear", "cross_attn.proj"):
                W[p + n + ".w"], W[p + n + ".b"] = mat(p + n + ".weight"), vec(p + n + ".bias")
            for n in ("attn.q_norm", "attn.k_norm", "cross_attn.q_norm", "cross_attn.k_norm"):
                W[p + n] = vec(p + n + ".weight")
            W[p + "ffn.w13"] = torch.cat([mat(p + "ffn.w1.weight"), mat(p + "ffn.w3.weight")], 0).contiguous()
            W[p + "ffn.w2"] = mat(p + "ffn.w2.weight")
        self.w = W
        return self

    def init_random(self, seed: int = 0):
        """Synthetic weights of the right shapes, generated on the device (there are no checkpoints offline)."""
        cfg, dev = self.cfg, self.device
        g = torch.Generator(device=dev).manual_seed(seed)
        C, Ct, Hd = cfg.hidden_size, cfg.adaln_tembed_dim, cfg.ffn_hidden

        def mat(n, k, std=None):
            std = std if std is not None else 1.0 / math.sqrt(k)
            return (torch.randn(n, k, generator=g, device=dev, dtype=torch.float32) * std).to(torch.bfloat16)

        def vec(n, std=0.02, base=0.0):
            return (torch.randn(n, generator=g, device=dev, dtype=torch.float32) * std + base).to(torch.bfloat16).float()

        W = {"patch.w": mat(C, cfg.in_channels * 4), "patch.b": vec(C)}
        for n, (o, i) in {"t_embedder.mlp.0": (Ct, cfg.frequency_embedding_size), "t_embedder.mlp.2": (Ct, Ct),
                          "y_embedder.y_proj.0": (C, cfg.caption_channels), "y_embedder.y_proj.2": (C, C),
                          "final_layer.linear": (4 * cfg.out_channels, C)}.items():
            W[n + ".w"], W[n + ".b"] = mat(o, i), vec(o)
        W["final_layer.adaLN_modulation.1.w"], W["final_layer.adaLN_modulation.1.b"] = mat(2 * C, Ct, 0.5 / math.sqrt(Ct)), vec(2 * C)
        W["ada.w"], W["ada.b"] = mat(cfg.depth * 6 * C, Ct, 0.5 / math.sqrt(Ct)), vec(cfg.depth * 6 * C)
        for i in range(cfg.depth):
            p = f"blocks.{i}."
            W[p + "norm.w"], W[p + "norm.b"] = vec(C, 0.05, 1.0), vec(C)
            for n, o in (("attn.qkv", 3 * C), ("attn.proj", C), ("cross_attn.q_linear", C), ("cross_attn.kv_linear", 2 * C),
                         ("cross_attn.proj", C)):
                W[p + n + ".w"], W[p + n + ".b"] = mat(o, C), vec(o)
            for n in ("attn.q_norm", "attn.k_norm", "cross_attn.q_norm", "cross_attn.k_norm"):
                W[p + n] = vec(128, 0.05, 1.0)
            W[p + "ffn.w13"], W[p + "ffn.w2"] = mat(2 * Hd, C), mat(C, Hd)
        self.w = W
        return self

    def param_bytes(self) -> int:
        return sum(t.numel() * t.element_size() for t in self.w.values())

    # ------------------------------------------------------------------------------------------------------------
    def _buf(self, name, shape, dtype, zero=False):
        key = (name, tuple(shape), dtype)
        t = self._ws.get(key)
        if t is None:
            t = (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=self.device)
            self._ws[key] = t
        return t

    def _rope_tables(self, f, h, w):
        key = (f, h, w)
        if key not in self._rope:
            c, s = rope_tables(128, f, h, w)
            self._rope[key] = (c.to(self.device), s.to(self.device))
        return self._rope[key]

    def _act(self, a, out_dtype, mode):
        out = torch.empty(a.shape, dtype=out_dtype, device=a.device)
        dt = {torch.float32: WF_F32, torch.bfloat16: WF_BF16}
        call("wf_act", a.data_ptr(), dt[a.dtype], None, 0, out.data_ptr(), dt[out_dtype], mode, a.numel(), ops.stream())
        return out

    def _gemm_f32(self, a: torch.Tensor, w, b, out):
        """out f32 = a f32 @ w^T + b with the fp32 activation split into a hi + lo bf16 pair (two MFMA GEMMs accumulating in fp32):
        what the reference's fp32 autocast regions compute from bf16-valued weights, to ~2^-17 relative."""
        hi = ops.cast(a, torch.bfloat16)
        lo = ops.cast(a - hi.float(), torch.bfloat16)
        gemm(hi, w, b, out, EPI_F32)
        gemm(lo, w, None, out, EPI_F32_ACC)
        return out

    def _ln(self, x, mul, add, mod_ld, rows_per_group, plus_one, out, row0=0, gidx=None):
        L, C = x.shape
        if L == 0:
            return
        call("wf_lc_ln_modulate", x.data_ptr(), mul.data_ptr(), add.data_ptr(), mod_ld, rows_per_group, row0,
             gidx.data_ptr() if gidx is not None else None, 1 if plus_one else 0, out.data_ptr(), L, C, float(self.cfg.eps), ops.stream())

    def _resid(self, x, y, gate, gate_ld, rows_per_group, row0=0, gidx=None):
        L, C = x.shape
        if L == 0:
            return
        call("wf_lc_gate_residual", x.data_ptr(), y.data_ptr(), y.stride(0), gate.data_ptr() if gate is not None else None, gate_ld,
             rows_per_group, row0, gidx.data_ptr() if gidx is not None and gate is not None else None, L, C, ops.stream())

    def _heads(self, src, col0, weight, cos, sin, out, r0, r1, out_scale=1.0):
        """Rows [r0, r1) of columns [col0, col0 + C) of src -> out [H, Lout, 128] rows [0, r1 - r0)."""
        if r1 <= r0:
            return
        view = src[r0:r1, col0:col0 + self.cfg.hidden_size]
        call("wf_lc_norm_heads", view.data_ptr(), src.stride(0), weight.data_ptr(),
             cos[r0:r1].data_ptr() if cos is not None else None, sin[r0:r1].data_ptr() if sin is not None else None,
             out.data_ptr(), r1 - r0, out.shape[1], self.cfg.num_heads, float(self.cfg.eps), float(out_scale), ops.stream())

    def _vt(self, src, col0, out, L):
        view = src[:, col0:col0 + self.cfg.hidden_size]
        call("wf_v_transpose", view.data_ptr(), src.stride(0), out.data_ptr(), L, out.shape[1] * 64, self.cfg.num_heads, ops.stream())

    # ------------------------------------------------------------------------------------------------------------
    # Exchange of a forward WITHOUT a lock-step partner (the distilled schedule has no CFG; see dit.WanTransformer3DModel for the modes);
    # the two samples of a CFG batch are advanced in lock-step with the one-event all-gather (forward_tokens_pair)
    exchange_mode = "chunked"
    exchange_chunks = 2
    pair_lockstep = True
    attn_prescale = True
    attn_track_max = False

    def _exchange(self, tag: str, H: int, shard_len: int, mode: str, chunks: int):
        from .parallel import KVExchange
        key = ("kvx" + tag, H, shard_len, mode, chunks, id(self.comm))
        ex = self._ws.get(key)
        if ex is None:
            ex = self._ws[key] = KVExchange(self.comm, H, shard_len, mode, chunks, self.device)
        return ex

    def forward_tokens(self, x_in: torch.Tensor, timesteps, caption: torch.Tensor, caption_mask: Optional[torch.Tensor] = None,
                       num_cond_latents: int = 0) -> torch.Tensor:
        """One sample.  x_in [16, T, Hh, Ww] bf16; timesteps: T host floats; caption [N, caption_channels] bf16; caption_mask [N]
        host / device ints (0 = padding) or None -> velocity [16, T, Hh, Ww] fp32  (LCD:279-366)."""
        out = [None]
        for _ in self._forward_steps(x_in, timesteps, caption, caption_mask, num_cond_latents, "", out, self.exchange_mode):
            pass
        return out[0]

    def forward_tokens_pair(self, sample_a, sample_b, num_cond_latents: int = 0):
        """The two samples of a CFG batch (pipeline_longcat_video.py:857-866: [negative, positive] concatenated on the batch axis; each
        sample = (x_in, timesteps, caption, caption_mask) as for forward_tokens).  The reference's batch is one tensor through one network
        call; here the two samples are two forwards advanced in LOCK-STEP, one layer apart, under sequence parallelism: while sample A's
        K / V^T all-gather of block i is in flight on the communication stream, sample B computes its block i-1 attention / FFN, and vice
        versa (the scheme of dit.WanTransformer3DModel.forward_tokens_pair).  Each sample issues exactly the kernels of forward_tokens in
        exchange mode "gather" on its own buffers: bit-identical to two sequential calls."""
        oa, ob = [None], [None]
        ga = self._forward_steps(*sample_a, num_cond_latents, "", oa, "gather")
        gb = self._forward_steps(*sample_b, num_cond_latents, "#b", ob, "gather")
        live = [ga, gb]
        while live:
            for gen in list(live):
                try:
                    next(gen)
                except StopIteration:
                    live.remove(gen)
        return oa[0], ob[0]

    def _forward_steps(self, x_in, timesteps, caption, caption_mask, num_cond_latents, tag, result, mode="gather"):
        """Generator over one forward: yields once per dense block, right after that block's K / V^T exchange has been launched (where
        another forward can usefully take over the compute stream); `tag` separates the workspaces of concurrent forwards; the velocity
        lands in result[0].

        With `comm` (one process per GPU) the tokens are split into contiguous shards, weights replicated: every rank runs the row-wise
        work on its shard, K and blocked V^T shards are exchanged once per block (parallel.KVExchange) and consumed in place by the
        attention kernel (segment addressing), the 64-column output rows are gathered at the end.  The condition / noise split of
        LCA:123-138 is by GLOBAL token index: a rank's rows below the first frame boundary are condition queries (keys < nc), the rest
        noise queries."""
        cfg, W, dev = self.cfg, self.w, self.device
        bf, f32 = torch.bfloat16, torch.float32
        Cin, T, Hh, Ww = x_in.shape
        assert Cin == cfg.in_channels and len(timesteps) == T
        C, H, Hd, Ct = cfg.hidden_size, cfg.num_heads, cfg.ffn_hidden, cfg.adaln_tembed_dim
        h2, w2 = Hh // 2, Ww // 2
        tpf = h2 * w2
        L, Lp = T * tpf, _pad64(T * tpf)
        nc = int(num_cond_latents or 0) * tpf
        assert 0 <= nc < L
        scale = 1.0 / math.sqrt(128.0)
        cos, sin = self._rope_tables(T, h2, w2)
        _buf = lambda name, shape, dtype, zero=False: self._buf(name + tag, shape, dtype, zero)  # noqa: E731
        comm = self.comm
        use_bsa = self._bsa and T > 1  # LCA:57: "bsa will not be used in image training / sampling"
        gidx = perm = pos = None
        blk = 64
        if use_bsa:
            # The refine pass keeps the WHOLE network in 3D-block token order (bsa_interface.py:600-604): every row-wise op is order
            # agnostic once the per-frame AdaLN kernels take a per-row frame index, the RoPE tables are permuted once, and the
            # velocity rows are put back in (T, H, W) order at the end -- no per-layer permutes, and contiguous runs of blocks are the
            # shards of the sequence-parallel job.  The first nc rows of the block order are the condition tokens.
            from . import bsa
            cq, ck = self.bsa_params["chunk_3d_shape_q"], self.bsa_params["chunk_3d_shape_k"]
            if list(cq) != list(ck):
                raise NotImplementedError("different query / key block shapes")
            ncl = int(num_cond_latents or 0)
            if ncl % cq[0] or (T - ncl) % cq[0]:
                raise ValueError(f"block-sparse attention needs the condition ({ncl}) and noise ({T - ncl}) latent frames to be "
                                 f"multiples of {cq[0]} (the reference pads them: pipeline_longcat_video.py:1417-1419)")
            perm, pos = bsa.block_permutation(T, h2, w2, cq, dev)
            blk = cq[0] * cq[1] * cq[2]
            key = ("blk", T, h2, w2, tuple(cq))
            if key not in self._rope:
                pl = perm.long()
                self._rope[key] = (cos[pl].contiguous(), sin[pl].contiguous(), (perm // tpf).to(torch.int32).contiguous())
            cos, sin, gidx = self._rope[key]
        if comm is not None:
            from .parallel import ShardPlan, gather_rows, shard_plan
            if use_bsa:  # whole 256-row query groups (two 128-token / four 64-token blocks) per rank
                per = (L + comm.world - 1) // comm.world
                plan = ShardPlan(L=L, P=comm.world, shard_len=(per + 255) // 256 * 256)
            else:
                plan = shard_plan(L, comm.world)
            lo, hi = plan.bounds(comm.rank)
            Lr, Sp = hi - lo, plan.shard_len
            if Lr <= 0:
                raise ValueError(f"sequence-parallel plan leaves rank {comm.rank} of {comm.world} without tokens ({L} tokens in shards "
                                 f"of {Sp}); use fewer ranks for this size")
        else:
            plan, lo, Lr, Sp = None, 0, L, Lp
        ncr = min(max(nc - lo, 0), Lr)  # this rank's condition rows
        cos, sin = cos[lo:lo + Lr], sin[lo:lo + Lr]

        # ---- embeddings ----
        tok = _buf("tok", (L, Cin * 4), bf)
        call("wf_patchify", x_in.data_ptr(), tok.data_ptr(), Cin, T, Hh, Ww, ops.stream())
        if use_bsa:
            tokb = _buf("tokb", (L, Cin * 4), bf)
            call("wf_gather_rows_bf16", tok.data_ptr(), tok.stride(0), perm.data_ptr(), tokb.data_ptr(), tokb.stride(0), L, Cin * 4,
                 ops.stream())
            tok = tokb
        tok = tok[lo:lo + Lr]
        L_all, nc_all = L, nc
        L, nc = Lr, ncr  # from here on L / nc are this rank's row counts; L_all / nc_all the key counts
        x = _buf("x", (L, C), bf)
        gemm(tok, W["patch.w"], W["patch.b"], x, EPI_BF16)  # LCB:112 (Conv3d with kernel = stride = patch)
        tf = timestep_embedding(timesteps, cfg.frequency_embedding_size).to(dev)  # LCB:201-206
        t0 = self._gemm_f32(tf, W["t_embedder.mlp.0.w"], W["t_embedder.mlp.0.b"], _buf("t0", (T, Ct), f32))
        t = self._gemm_f32(self._act(t0, f32, 0), W["t_embedder.mlp.2.w"], W["t_embedder.mlp.2.b"], _buf("t", (T, Ct), f32))
        st = self._act(t, f32, 0)  # SiLU(t), shared by every adaLN_modulation (LCD:40-43, LCB:156)
        ada = self._gemm_f32(st, W["ada.w"], W["ada.b"], _buf("ada", (T, W["ada.w"].shape[0]), f32))   # (all stored blocks: a run on the first cfg.depth blocks only reads its own columns)
        fmod = self._gemm_f32(st, W["final_layer.adaLN_modulation.1.w"], W["final_layer.adaLN_modulation.1.b"], _buf("fmod", (T, 2 * C), f32))
        # caption: Linear -> GELU(tanh) -> Linear (LCB:225-228), valid tokens only (LCD:319-325)
        cap = caption
        if caption_mask is not None:
            keep = torch.as_tensor(caption_mask).reshape(-1).to("cpu") != 0
            if not cfg.text_tokens_zero_pad:
                cap = caption[keep.to(caption.device)]
        n_txt = cap.shape[0]
        assert n_txt > 0, "empty caption"
        yh = _buf("yh", (n_txt, C), bf)
        gemm(cap.contiguous(), W["y_embedder.y_proj.0.w"], W["y_embedder.y_proj.0.b"], yh, EPI_BF16_GELU)
        y = _buf("y", (n_txt, C), bf)
        gemm(yh, W["y_embedder.y_proj.2.w"], W["y_embedder.y_proj.2.b"], y, EPI_BF16)
        if caption_mask is not None and cfg.text_tokens_zero_pad:
            y[(~keep).to(dev)] = 0  # LCD:315-317
        Ltp = _pad64(n_txt)

        hbuf = _buf("h", (L, C), bf)
        qkv = _buf("qkv", (L, 3 * C), bf)
        qh_c = _buf("qh_c", (H, max(nc, 1), 128), bf)
        qh_n = _buf("qh_n", (H, max(L - nc, 1), 128), bf)  # (a rank of a sequence-parallel job may hold condition rows only)
        # dense self-attention (no block gating on Q): as in the Wan DiT (dit.py), softmax_scale * log2(e) is folded into Q by its producer and
        # the kernel runs its exp2-domain form (softmax_scale = 0), without max tracking where the per-head norm bound allows it.  The
        # block-sparse pass keeps the in-kernel scale: its Q also feeds the gating.
        prescale = (not use_bsa) and bool(self.attn_prescale)
        q_scale, sa_scale = (scale * 1.4426950408889634, 0.0) if prescale else (1.0, scale)
        use_bounds = prescale and not self.attn_track_max
        qm_c = _buf("qmax2_c", (H,), f32) if use_bounds else None
        qm_n = _buf("qmax2_n", (H,), f32) if use_bounds else None
        ex = kh = vt = km = None
        if comm is not None and not use_bsa:
            # dense blocks: K, V^T and the norm bounds of this rank's shard go straight into its slot of the exchange buffers
            if not prescale or comm.world == 1:
                mode = "gather"   # part launches are built for the pre-scaled-Q form
            ex = self._exchange(tag, H, Sp, mode, int(self.exchange_chunks))
        else:
            kh = _buf("kh", (H, Sp, 128), bf, zero=True)
            vt = _buf("vt", (H, Sp // 64, 128, 64), bf)
            km = _buf("kmax2", (H,), f32) if use_bounds else None
            if comm is not None:  # block-sparse blocks gather K / V^T densely (the sparse kernel addresses [P][H][S][128])
                kh_all = _buf("kh_all", (comm.world, H, Sp, 128), bf, zero=True)
                vt_all = _buf("vt_all", (comm.world, H, Sp // 64, 128, 64), bf)
        ao = _buf("ao", (L, C), bf)
        ys = _buf("ys", (L, C), bf)
        qc = _buf("qc", (L, C), bf)
        kvt = _buf("kvt", (n_txt, 2 * C), bf)
        kth = _buf("kth", (H, Ltp, 128), bf, zero=True)
        vtt = _buf("vtt", (H, Ltp // 64, 128, 64), bf)
        ffh = _buf("ffh", (L, 2 * Hd), bf)
        ffg = _buf("ffg", (L, Hd), bf)
        ald = ada.stride(0)
        if use_bsa:
            sparsity = self.bsa_params.get("sparsity")
            cdf_thr = self.bsa_params.get("cdf_threshold")
            if sparsity is None and cdf_thr is None:
                raise ValueError("bsa_params needs sparsity and / or cdf_threshold (bsa_interface.py:265-274)")

            def select(scores):  # bsa_interface.py:265-274 -> (block indices, per-row counts or None)
                if cdf_thr is None:
                    return bsa.select_topk(scores, float(sparsity)), None
                return bsa.select_cdf(scores, float(cdf_thr), None if sparsity is None else float(sparsity))

            fused_sel = not os.environ.get("WF_BSA_TORCH_SELECT")  # (the env switch keeps the torch selection paths testable)
            self.last_bsa_indices = []

        # sequence-parallel jobs: the caption K / V^T of layer i are computed by rank i (mod P) only and all-gathered once per forward (see
        # dit.py: work that does not shrink with the token shard)
        ctx_shared = ctx_events = None
        if comm is not None and comm.world > 1:
            P_, nl = comm.world, (cfg.depth + comm.world - 1) // comm.world
            loc = [_buf("ckv_loc0", (nl, H, Ltp, 128), bf, zero=True), _buf("ckv_loc1", (nl, H, Ltp // 64, 128, 64), bf)]
            allb = [_buf(f"ckv_all{j}", (P_,) + tuple(t.shape), bf) for j, t in enumerate(loc)]
            for j in range(nl):
                i = comm.rank + P_ * j
                if i >= cfg.depth:
                    break
                p = f"blocks.{i}."
                gemm(y, W[p + "cross_attn.kv_linear.w"], W[p + "cross_attn.kv_linear.b"], kvt, EPI_BF16)
                self._heads(kvt, 0, W[p + "cross_attn.k_norm"], None, None, loc[0][j], 0, n_txt)
                self._vt(kvt, C, loc[1][j], n_txt)
            ctx_events = [comm.all_gather_async(a_, l_) for a_, l_ in zip(allb, loc)]
            ctx_shared = (allb, P_)

        for i in range(cfg.depth):
            p = f"blocks.{i}."
            m = ada[:, i * 6 * C:(i + 1) * 6 * C]
            shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = (m[:, j * C:(j + 1) * C] for j in range(6))
            # ---- self-attention (LCD:91-104, LCA:105-145) ----
            self._ln(x, scale_msa, shift_msa, ald, tpf, True, hbuf, row0=lo, gidx=gidx)
            gemm(hbuf, W[p + "attn.qkv.w"], W[p + "attn.qkv.b"], qkv, EPI_BF16)
            if use_bsa:
                # LCA:57-66 + bsa_interface.py:612-659 on rows that already are in block order: gating = mean-pooled q / k blocks ->
                # bf16 block scores -> top-k / cdf selection per query block -> sparse attention over the selected key blocks
                self._heads(qkv, 0, W[p + "attn.q_norm"], cos, sin, qh_c, 0, nc)
                self._heads(qkv, 0, W[p + "attn.q_norm"], cos, sin, qh_n, nc, L)
                self._heads(qkv, C, W[p + "attn.k_norm"], cos, sin, kh, 0, L)
                self._vt(qkv, 2 * C, vt, L)
                kcmp = bsa.mean_pool(kh, blk)  # this rank's key blocks (zero rows past the last token pool to zero blocks)
                kk, vv = kh, vt
                if comm is not None:
                    evs = (comm.all_gather_async(kh_all, kh), comm.all_gather_async(vt_all, vt))
                    kc_all = torch.empty((comm.world,) + tuple(kcmp.shape), dtype=bf, device=dev)
                    comm.all_gather(kc_all, kcmp)
                    kcmp = kc_all.permute(1, 0, 2, 3).reshape(H, -1, 128)
                    for ev in evs:
                        if ev is not None:
                            torch.cuda.current_stream().wait_event(ev)
                    kk, vv = kh_all, vt_all
                kcmp = kcmp[:, :L_all // blk].contiguous()
                picked = []
                for nrows, qrows, orows, nkb in ((nc, qh_c, ao[:nc], nc_all // blk), (L - nc, qh_n, ao[nc:], L_all // blk)):
                    if nrows == 0:
                        continue
                    sc = bsa.block_scores(bsa.mean_pool(qrows, blk), kcmp if nkb == L_all // blk else kcmp[:, :nkb].contiguous())
                    if fused_sel and nkb <= bsa.TOPK_MAX_BLOCKS and cdf_thr is None:  # selection + list building in one kernel
                        picked.append(bsa.sparse_attention_topk(qrows, kk, vv, orows, sc, float(sparsity), scale, blk))
                    elif fused_sel and nkb <= bsa.TOPK_MAX_BLOCKS:  # the cdf rule: counts (sort + scan in LDS), then the same list kernel
                        picked.append(bsa.sparse_attention_cdf(qrows, kk, vv, orows, sc, float(cdf_thr),
                                                               None if sparsity is None else float(sparsity), scale, blk))
                    else:
                        idx, lens = select(sc)
                        bsa.sparse_attention(qrows, kk, vv, orows, idx, scale, nkb, lens, blk)
                        picked.append(idx if lens is None else (idx, lens))
                self.last_bsa_indices.append(picked)
            else:
                self._heads(qkv, 0, W[p + "attn.q_norm"], cos, sin, qh_c, 0, nc, out_scale=q_scale)
                self._heads(qkv, 0, W[p + "attn.q_norm"], cos, sin, qh_n, nc, L, out_scale=q_scale)
                if ex is None:
                    self._heads(qkv, C, W[p + "attn.k_norm"], cos, sin, kh, 0, L)
                    self._vt(qkv, 2 * C, vt, L)
                    if use_bounds:  # zero rows past L do not raise a maximum: the whole (padded) shard is scanned
                        head_max_norm2(kh, Sp, km)
                else:
                    for g in range(ex.G):
                        r0, r1 = ex.chunk_rows(g, L)
                        if r1 > r0:
                            self._heads(qkv, C, W[p + "attn.k_norm"], cos, sin, ex.own_k(g), r0, r1)
                            self._vt(qkv[r0:r1], 2 * C, ex.own_vt(g), r1 - r0)
                            if use_bounds:
                                head_max_norm2(ex.own_k(g), ex.chunk_len(g), ex.own_km(g))
                    ex.launch()
                    yield i
                if use_bounds and nc > 0:
                    head_max_norm2(qh_c, nc, qm_c)
                if use_bounds and L - nc > 0:
                    head_max_norm2(qh_n, L - nc, qm_n)
                if ex is None:
                    if nc > 0:  # condition tokens see condition tokens only (LCA:127-131)
                        attention(qh_c, kh, vt, ao[:nc], nc_all, sa_scale, kmax2=km, qmax2=qm_c)
                    if L - nc > 0:  # noise tokens see everything (LCA:133-134)
                        attention(qh_n, kh, vt, ao[nc:], L_all, sa_scale, profile=True, kmax2=km, qmax2=qm_n)
                else:
                    # the noise queries first: their sweep is what the exchange hides under; the condition queries see the first nc_all
                    # keys only (rank 0's first rows), their windows are ready by then
                    if L - nc > 0:
                        attention_exchange(qh_n, ex, ao[nc:], L_all, sa_scale, qm_n, use_bounds=use_bounds, profile=True, release=False)
                    if nc > 0:
                        attention_exchange(qh_c, ex, ao[:nc], nc_all, sa_scale, qm_c, use_bounds=use_bounds, release=False,
                                           comm_profile=L - nc <= 0)
                    ex.wait_all()
            gemm(ao, W[p + "attn.proj.w"], W[p + "attn.proj.b"], ys, EPI_BF16)
            self._resid(x, ys, gate_msa, ald, tpf, row0=lo, gidx=gidx)
            # ---- cross-attention on the noise tokens (LCD:108-111, LCA:218-276) ----
            if L - nc > 0:
                self._ln(x[nc:], W[p + "norm.w"], W[p + "norm.b"], 0, 0, False, hbuf[nc:])
                gemm(hbuf[nc:], W[p + "cross_attn.q_linear.w"], W[p + "cross_attn.q_linear.b"], qc[nc:], EPI_BF16)
                self._heads(qc, 0, W[p + "cross_attn.q_norm"], None, None, qh_n, nc, L)
                if ctx_shared is not None:
                    if ctx_events is not None:
                        for ev in ctx_events:
                            if ev is not None:
                                torch.cuda.current_stream().wait_event(ev)
                        ctx_events = None
                    kth_i, vtt_i = (a_[i % ctx_shared[1], i // ctx_shared[1]] for a_ in ctx_shared[0])
                else:
                    gemm(y, W[p + "cross_attn.kv_linear.w"], W[p + "cross_attn.kv_linear.b"], kvt, EPI_BF16)
                    self._heads(kvt, 0, W[p + "cross_attn.k_norm"], None, None, kth, 0, n_txt)
                    self._vt(kvt, C, vtt, n_txt)
                    kth_i, vtt_i = kth, vtt
                attention(qh_n, kth_i, vtt_i, ao[nc:], n_txt, scale)
                gemm(ao[nc:], W[p + "cross_attn.proj.w"], W[p + "cross_attn.proj.b"], ys[nc:], EPI_BF16)
                self._resid(x[nc:], ys[nc:], None, 0, 0)
            # ---- SwiGLU FFN (LCD:113-120, LCB:36-37) ----
            self._ln(x, scale_mlp, shift_mlp, ald, tpf, True, hbuf, row0=lo, gidx=gidx)
            gemm(hbuf, W[p + "ffn.w13"], None, ffh, EPI_BF16)
            call("wf_lc_swiglu", ffh.data_ptr(), ffh.stride(0), ffg.data_ptr(), L, Hd, ops.stream())
            gemm(ffg, W[p + "ffn.w2"], None, ys, EPI_BF16)
            self._resid(x, ys, gate_mlp, ald, tpf, row0=lo, gidx=gidx)

        # ---- final layer (LCB:159-168) + unpatchify (LCD:371-392) ----
        self._ln(x, fmod[:, C:], fmod[:, :C], fmod.stride(0), tpf, True, hbuf, row0=lo, gidx=gidx)
        yo = _buf("yo", (L, 4 * cfg.out_channels), f32)
        gemm(hbuf, W["final_layer.linear.w"], W["final_layer.linear.b"], yo, EPI_F32)
        if comm is not None:
            yo = gather_rows(comm, yo, plan).contiguous()
        if use_bsa:  # velocity rows back to (T, H, W) order: bsa_interface.py:606-610 (fp32 rows moved as 16-byte chunks)
            yt = torch.empty_like(yo)
            call("wf_gather_rows_bf16", yo.data_ptr(), 2 * yo.stride(0), pos.data_ptr(), yt.data_ptr(), 2 * yt.stride(0), L_all,
                 2 * yo.shape[1], ops.stream())
            yo = yt
        out = torch.empty((cfg.out_channels, T, Hh, Ww), dtype=f32, device=dev)
        call("wf_unpatchify", yo.data_ptr(), out.data_ptr(), cfg.out_channels, T, Hh, Ww, ops.stream())
        result[0] = out

    def __call__(self, hidden_states: torch.Tensor, timestep: torch.Tensor, encoder_hidden_states: torch.Tensor,
                 encoder_attention_mask: Optional[torch.Tensor] = None, num_cond_latents: int = 0, return_kv: bool = False,
                 kv_cache_dict=None, skip_crs_attn: bool = False, offload_kv_cache: bool = False) -> torch.Tensor:
        if return_kv or kv_cache_dict or skip_crs_attn:
            raise NotImplementedError("KV-cache continuation (LCA:147-181) is not built; the guided i2v path does not use it")
        B, _, T, _, _ = hidden_states.shape
        ts = torch.as_tensor(timestep)
        if ts.dim() == 1:
            ts = ts.unsqueeze(1).expand(-1, T)  # LCD:299-301
        # LCD:304-306: the reference casts the timesteps to the model dtype (bf16) before embedding them
        ts = ts.to(self.dtype).float().cpu()
        cap = encoder_hidden_states
        if cap.dim() == 4:
            cap = cap[:, 0]
        samples = []
        for b in range(B):
            x = hidden_states[b]
            if x.dtype != torch.bfloat16:
                x = ops.cast(x.contiguous(), torch.bfloat16)
            mask = encoder_attention_mask[b] if encoder_attention_mask is not None else None
            samples.append((x.contiguous(), ts[b].tolist(), cap[b].to(torch.bfloat16).contiguous(), mask))
        if B == 2 and self.comm is not None and self.comm.world > 1 and self.pair_lockstep:
            # the CFG batch (pipeline_longcat_video.py:857-866): two forwards in lock-step, each exchange hidden under the other's block
            return torch.stack(self.forward_tokens_pair(samples[0], samples[1], num_cond_latents))
        return torch.stack([self.forward_tokens(*smp, num_cond_latents) for smp in samples])
