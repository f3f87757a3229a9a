"""Wan2.1 image-to-video DiT (WanTransformer3DModel) on hand-written HIP kernels.

Speaks the diffusers call protocol the reference sampler uses (PIPE:593-600):
    transformer(hidden_states[B,36,T,h,w], timestep[B], encoder_hidden_states[B,512,4096],
                encoder_hidden_states_image[B,257,1280], attention_kwargs=None, return_dict=False)[0] -> [B,16,T,h,w]
The arithmetic follows the in-tree statement of the model, /root/reference/wan_for_worldforge/wan/modules/model.py
(the executed class lives in diffusers, outside the reference tree); file:line citations are on the kernels
(csrc/gemm.hip, attention.hip, dit_ops.hip) and below.  Every FLOP runs in libwf_hip.so; PyTorch only owns the buffers.

Numerics: bf16 weights and bf16 GEMM / attention operands with fp32 accumulation, fp32 residual stream, fp32
LayerNorm / modulation (model.py:296-313) -- the reference's bf16 autocast flow.

Weights are keyed like the in-tree twin's state_dict ("blocks.3.self_attn.q.weight"); `diffusers_key_map` translates a
diffusers checkpoint.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass
from types import SimpleNamespace
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import _ffi, ops
from ._ffi import WF_BF16, WF_F32, call

EPI_BF16, EPI_BF16_GELU, EPI_F32, EPI_RESID, EPI_F32_ACC = 0, 1, 2, 3, 4


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
    patch_size: Tuple[int, int, int] = (1, 2, 2)
    eps: float = 1e-6

    @staticmethod
    def wan_i2v_14b() -> "DiTConfig":
        """wan/configs/wan_i2v_14B.py:27-36."""
        return DiTConfig()


def _pad64(n: int) -> int:
    return (n + 63) // 64 * 64


def gemm(x: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], out: torch.Tensor, epi: int,
         gate: Optional[torch.Tensor] = None):
    """out[M,N] = epi(x[M,K] @ w[N,K]^T + bias)."""
    M, K = x.shape
    N = w.shape[0]
    assert w.shape[1] == K and x.dtype == torch.bfloat16 and w.dtype == torch.bfloat16
    assert out.shape[0] == M and out.shape[1] == N
    call("wf_gemm_bf16", x.data_ptr(), w.data_ptr(), bias.data_ptr() if bias is not None else None, out.data_ptr(),
         gate.data_ptr() if gate is not None else None, M, N, K, x.stride(0), w.stride(0), out.stride(0), epi, ops.stream())
    return out


def ffn_padded_features(f: int) -> int:
    """Rows the FFN-up weight is stored with: the next multiple of 320 when that costs at most 3 % more features (13 824 -> 14 080).
    `wf_gemm_bf16` runs its 320-feature tile only on N % 320 == 0 (fewer fragment reads per MFMA, and 2.75 instead of 3.4 rounds of 256
    workgroups at the 8-rank token count); the padding rows are zero (weight and bias), so the padded hidden columns are gelu(0) = 0,
    FFN-down reads the first `f` columns through the row stride, and every real element is the same K-ordered accumulation as before
    (the result does not depend on the tile width: tests/test_gpu_dit.py)."""
    fp = -(-f // 320) * 320
    return fp if (fp - f) * 100 <= 3 * f else f


def _pad_rows(t: torch.Tensor, rows: int) -> torch.Tensor:
    if t.shape[0] == rows:
        return t
    pad = torch.zeros((rows - t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    return torch.cat([t, pad], 0).contiguous()


PROFILE_ATTN = None  # bench.py sets this to a list: (start, end) HIP events around every self-attention launch
PROFILE_COMM = None  # bench.py (N > 1) sets this to a list; per layer one (start, end) HIP event pair around the compute stream's wait for the
                     # K / V^T exchange -- or, for an own-first sweep (attention_exchange, modes chunked / bcast), the LIST of such pairs of the layer


def comm_wait_ms(entry) -> float:
    """Exposed wait of one PROFILE_COMM entry in ms (after a device synchronise)."""
    if isinstance(entry, list):
        return sum(a.elapsed_time(b) for a, b in entry)
    return entry[0].elapsed_time(entry[1])


def kv_splits(H: int, Lq: int, kv_len: int, n_cu: int = 256) -> int:
    """KV splits for the attention launch: the kernel runs one 256-row workgroup per CU, so a launch of W = ceil(Lq/256) * H workgroups
    takes ceil(W / n_cu) rounds; when Lq is short (one rank's token shard of the sequence-parallel DiT: W = 640 at 8 ranks = 2.5
    rounds) splitting the KV sweep in two fills the last round.  A function of the shapes only (never of timing)."""
    if kv_len < 128 * 64:
        return 1
    w = -(-Lq // 256) * H
    eff = lambda n: (w * n / n_cu) / -(-(w * n) // n_cu)
    return 2 if eff(2) > eff(1) + 0.08 else 1


def head_max_norm2(k: torch.Tensor, L: int, out: torch.Tensor) -> torch.Tensor:
    """k [H, Lp, 128] bf16 -> out [H] f32 = max over the first L rows of |k|^2 per head (see wf_head_max_norm2)."""
    H, Lp, _ = k.shape
    out.zero_()
    call("wf_head_max_norm2", k.data_ptr(), H, L, Lp, out.data_ptr(), ops.stream())
    return out


def attention(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, out: torch.Tensor, kv_len: int, scale: float,
              accumulate: bool = False, profile: bool = False, nsplit: Optional[int] = None, kmax2: Optional[torch.Tensor] = None,
              qmax2: Optional[torch.Tensor] = None):
    """q [H,Lq,128]; k [H,Lkp,128] and vt [H,Lkp/64,128,64], or their all-gathered per-rank shards k [P,H,S,128],
    vt [P,H,S/64,128,64] (S = shard length, keys in shard-major order; dense, or the strided views of a packed exchange buffer:
    parallel.KVExchange, same slot stride for both) -> out [Lq, H*128] bf16."""
    H, Lq, D = q.shape
    seg_stride = 0
    if k.dim() == 4:
        P, _, seg, _ = k.shape
        assert vt.shape == (P, H, seg // 64, 128, 64)
        assert k[0].is_contiguous() and vt[0].is_contiguous() and k.stride(0) == vt.stride(0)
        Lkp = P * seg
        seg_stride = 2 * k.stride(0)
    else:
        Lkp = seg = k.shape[1]
        assert vt.shape == (H, Lkp // 64, 128, 64)
    assert D == 128
    prof = PROFILE_ATTN if profile else None
    if nsplit is None:
        nsplit = kv_splits(H, Lq, kv_len)
    kmp, kmn, kms, qmp, qmn = None, 0, 0, None, 0
    if kmax2 is not None and qmax2 is not None:  # [H] or [P, H] f32 each (one vector per gathered shard); only with scale == 0 (pre-scaled Q)
        for t in (kmax2, qmax2):
            assert t.dtype == torch.float32 and t.stride(-1) == 1 and t.shape[-1] == H
        assert qmax2.is_contiguous()
        kmp, kmn, qmp, qmn = kmax2.data_ptr(), kmax2.numel() // H, qmax2.data_ptr(), qmax2.numel() // H
        kms = kmax2.stride(0) if kmax2.dim() == 2 else 0
    if prof is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    if nsplit > 1:
        ws = ops._workspace("attn_split", (_ffi.lib().wf_attn_split_workspace_bytes(H, Lq, nsplit) + 3) // 4, q.device)
        call("wf_attn_fwd_split", q.data_ptr(), k.data_ptr(), vt.data_ptr(), out.data_ptr(), H, Lq, Lkp, kv_len, seg, seg_stride, out.stride(0),
             float(scale), 1 if accumulate else 0, nsplit, ws.data_ptr(), kmp, kmn, kms, qmp, qmn, ops.stream())
    else:
        call("wf_attn_fwd", q.data_ptr(), k.data_ptr(), vt.data_ptr(), out.data_ptr(), H, Lq, Lkp, kv_len, seg, seg_stride, out.stride(0),
             float(scale), 1 if accumulate else 0, kmp, kmn, kms, qmp, qmn, ops.stream())
    if prof is not None:
        ev1.record()
        prof.append((ev0, ev1))
    return out


def attention_exchange(q: torch.Tensor, ex, out: torch.Tensor, kv_len: int, scale: float, qmax2: Optional[torch.Tensor],
                       use_bounds: bool = True, profile: bool = False, peer_groups: int = 2, release: bool = True,
                       comm_profile: bool = True):
    """Self-attention of one sequence-parallel layer over the exchange buffers `ex` (parallel.KVExchange, `launch`ed earlier): q [H,Lq,128]
    attends to the first kv_len keys of the sequence -> out [Lq, H*128] bf16.
      gather            wait for the one event, ONE launch over the P segments (split-KV when the query shard is short): the tiles of
                        the single-GPU sweep in the same order.
      chunked / bcast   own-first part launches (parallel.sweep_plan; wf_attn_fwd_part needs the pre-scaled Q form, scale == 0): the
                        rank's own keys without any wait, then each window after the event of its chunk / last source -- the exposed
                        part of the exchange is what the compute stream still has to wait for THEN (PROFILE_COMM records every wait)
                        -- and one exact merge, in the last part launch's own epilogue (or wf_attn_merge when that launch has several
                        splits).  Result = the one-launch sweep up to the re-association of the fp32
                        partial sums (the class the 8-rank split sweep already has).
    release: the compute stream ends up behind the LAST collective (KVExchange.wait_all) before the caller goes on to write the
    exchange buffers again."""
    from .parallel import sweep_plan
    H, Lq, _ = q.shape
    cprof = PROFILE_COMM if comm_profile else None
    waits = []

    def timed_wait(i):
        if cprof is not None:
            cw0, cw1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            cw0.record()
        ex.wait(i)
        if cprof is not None:
            cw1.record()
            waits.append((cw0, cw1))

    bounds = use_bounds and qmax2 is not None
    if ex.mode == "gather":
        timed_wait(0)
        attention(q, ex.k[0], ex.vt[0], out, kv_len, scale, profile=profile, kmax2=ex.km[0] if bounds else None, qmax2=qmax2 if bounds else None)
        if cprof is not None:
            cprof.append(waits[0])
        return out
    assert scale == 0.0, "own-first sweeps are part launches: pre-scaled Q only"
    steps, nparts = sweep_plan(ex, kv_len, -(-Lq // 256) * H, peer_groups)
    prof = PROFILE_ATTN if profile else None
    if prof is not None:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    if nparts == 1:  # one window holds every key these queries see (LongCat's condition rows): a plain launch on that chunk's buffer
        st = steps[0]
        if st["wait"] is not None:
            timed_wait(st["wait"])
        g = st["chunk"]
        a, n = st["km"]
        attention(q, ex.k[g], ex.vt[g], out, ex.chunk_kv_len(kv_len, g), scale, nsplit=1, kmax2=ex.km[g][a:a + n] if bounds else None,
                  qmax2=qmax2 if bounds else None)
    else:
        from .parallel import MAX_ATTN_PARTS
        assert 2 <= nparts <= MAX_ATTN_PARTS, nparts
        ws = ops._workspace("attn_split", (_ffi.lib().wf_attn_split_workspace_bytes(H, Lq, nparts) + 3) // 4, q.device)
        waited = set()
        for st in steps:
            g = st["chunk"]
            if st["wait"] is not None and st["wait"] not in waited:
                timed_wait(st["wait"])
                waited.update(range(st["wait"] + 1))  # stream order: the earlier collectives have finished too
            sc = ex.chunk_len(g)
            kmp, kmn, kms, qmp, qmn = None, 0, 0, None, 0
            if bounds:
                a, n = st["km"]
                kmp, kmn, kms, qmp, qmn = ex.km[g][a].data_ptr(), n, ex.km_stride(g), qmax2.data_ptr(), qmax2.numel() // H
            (t0, t1, inner), w2 = st["win"], st["win2"] or (0, 0, 0)
            call("wf_attn_fwd_part", q.data_ptr(), ex.k[g].data_ptr(), ex.vt[g].data_ptr(), H, Lq, ex.P * sc, ex.chunk_kv_len(kv_len, g), sc,
                 ex.seg_stride_bytes(g), t0, t1, w2[0], w2[1], inner, st["slot"], nparts, ws.data_ptr(),
                 out.data_ptr() if st["merge"] else None, out.stride(0), kmp, kmn, kms, qmp, qmn, ops.stream())
        if not steps[-1]["merge"]:
            call("wf_attn_merge", out.data_ptr(), H, Lq, out.stride(0), 0, nparts, ws.data_ptr(), ops.stream())
    if prof is not None:
        ev1.record()
        prof.append((ev0, ev1))
    if release:
        ex.wait_all()
    if cprof is not None:
        cprof.append(waits)   # one entry per layer: the list of (start, end) event pairs of its waits
    return out


def cross_attention2(q: torch.Tensor, kc: torch.Tensor, vtc: torch.Tensor, out: torch.Tensor, Lk1p: int, kv_len1: int, kv_len2: int, scale: float):
    """WanI2VCrossAttention's two attentions (model.py:220-227) in one launch (wf_attn_cross2_fwd): q [H,Lq,128]; kc [H, Lk1p + Lk2p, 128] =
    context 1 (kv_len1 valid rows, zero-padded to Lk1p) then context 2; vtc [H, (Lk1p + Lk2p) / 64, 128, 64] -> out [Lq, H*128] bf16."""
    H, Lq, D = q.shape
    Lkp = kc.shape[1]
    assert D == 128 and kc.shape == (H, Lkp, 128) and vtc.shape == (H, Lkp // 64, 128, 64) and kc.is_contiguous() and vtc.is_contiguous()
    call("wf_attn_cross2_fwd", q.data_ptr(), kc.data_ptr(), vtc.data_ptr(), out.data_ptr(), H, Lq, Lk1p, kv_len1, Lkp - Lk1p, kv_len2,
         out.stride(0), float(scale), ops.stream())
    return out


def rope_tables(head_dim: int, f: int, h: int, w: int, theta: float = 10000.0):
    """model.py:32-39 + 478-485 + 57-62: cos/sin [f*h*w, head_dim/2], computed in fp64 on the host, stored fp32."""
    c = head_dim // 2
    d_hw = c // 3
    d_f = c - 2 * d_hw

    def axis(n, npairs):
        dim = 2 * npairs
        inv = 1.0 / np.power(theta, np.arange(0, dim, 2, dtype=np.float64) / dim)
        return np.outer(np.arange(n, dtype=np.float64), inv)

    af, ah, aw = axis(f, d_f), axis(h, d_hw), axis(w, d_hw)
    ang = np.concatenate([np.broadcast_to(af[:, None, None, :], (f, h, w, d_f)),
                          np.broadcast_to(ah[None, :, None, :], (f, h, w, d_hw)),
                          np.broadcast_to(aw[None, None, :, :], (f, h, w, d_hw))], axis=-1).reshape(f * h * w, c)
    return torch.from_numpy(np.cos(ang).astype(np.float32)), torch.from_numpy(np.sin(ang).astype(np.float32))


def sinusoidal_embedding_1d(dim: int, t: float) -> torch.Tensor:
    """model.py:18-28 (fp64 on the host) -> [1, dim] fp32."""
    half = dim // 2
    pos = np.float64(t)
    s = pos * np.power(10000.0, -np.arange(half, dtype=np.float64) / half)
    return torch.from_numpy(np.concatenate([np.cos(s), np.sin(s)])[None].astype(np.float32))


def diffusers_key_map(num_layers: int) -> Dict[str, str]:
    """diffusers WanTransformer3DModel parameter name -> in-tree twin name (verify on first contact
    with a real checkpoint's *.safetensors.index.json)."""
    m = {
        "patch_embedding": "patch_embedding",
        "condition_embedder.time_embedder.linear_1": "time_embedding.0",
        "condition_embedder.time_embedder.linear_2": "time_embedding.2",
        "condition_embedder.time_proj": "time_projection.1",
        "condition_embedder.text_embedder.linear_1": "text_embedding.0",
        "condition_embedder.text_embedder.linear_2": "text_embedding.2",
        "condition_embedder.image_embedder.norm1": "img_emb.proj.0",
        "condition_embedder.image_embedder.ff.net.0.proj": "img_emb.proj.1",
        "condition_embedder.image_embedder.ff.net.2": "img_emb.proj.3",
        "condition_embedder.image_embedder.norm2": "img_emb.proj.4",
        "proj_out": "head.head",
    }
    for i in range(num_layers):
        b, t = f"blocks.{i}.", f"blocks.{i}."
        for a, n in (("attn1", "self_attn"), ("attn2", "cross_attn")):
            for s, d in (("to_q", "q"), ("to_k", "k"), ("to_v", "v"), ("to_out.0", "o"), ("norm_q", "norm_q"),
                         ("norm_k", "norm_k")):
                m[f"{b}{a}.{s}"] = f"{t}{n}.{d}"
        m[f"{b}attn2.add_k_proj"] = f"{t}cross_attn.k_img"
        m[f"{b}attn2.add_v_proj"] = f"{t}cross_attn.v_img"
        m[f"{b}attn2.norm_added_k"] = f"{t}cross_attn.norm_k_img"
        m[f"{b}norm2"] = f"{t}norm3"
        m[f"{b}ffn.net.0.proj"] = f"{t}ffn.0"
        m[f"{b}ffn.net.2"] = f"{t}ffn.2"
    return m


class WanTransformer3DModel:
    dtype = torch.bfloat16

    def __init__(self, cfg: DiTConfig, device="cuda:0", comm=None):
        self.cfg = cfg
        self.comm = comm  # parallel.Comm for sequence parallelism (None = single GPU)
        self.config = SimpleNamespace(patch_size=cfg.patch_size, in_channels=cfg.in_dim, out_channels=cfg.out_dim)
        self.device = torch.device(device)
        self.w: Dict[str, torch.Tensor] = {}
        self._ws = {}
        self._rope = {}
        assert cfg.dim % cfg.num_heads == 0 and cfg.dim // cfg.num_heads == 128, "attention kernel is built for head_dim 128"

    # ------------------------------------------------------------------------------------------------------------
    # weights
    # ------------------------------------------------------------------------------------------------------------
    @property
    def w(self) -> Dict[str, torch.Tensor]:
        return self._w

    @w.setter
    def w(self, W: Dict[str, torch.Tensor]):
        """Every assignment of the weight dict (load_state_dict, init_random, sharing another instance's weights) starts a new weights
        version and drops what was derived from the old one (the prompt-context K / V cache)."""
        self._w = W
        self.weights_changed()

    def weights_changed(self):
        """Call after editing weight tensors IN PLACE (e.g. folding a LoRA into the same tensors): the prompt-context K / V cache is keyed by
        a monotonically increasing weights version, never by id() of a dict that may have been freed and its address reused."""
        self._wver = getattr(self, "_wver", 0) + 1
        self.__dict__.pop("_ctx_cache", None)

    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        """Twin-keyed state dict (any dtype / device) -> device tensors: matrices bf16, vectors fp32, q/k/v fused."""
        cfg, dev = self.cfg, self.device
        W = {}

        def mat(k):
            return sd[k].to(device=dev, dtype=torch.bfloat16).contiguous()

        def vec(k):
            # biases / norm weights / modulation: kept in fp32 on the device (the upcast of a bf16 checkpoint value is exact)
            return sd[k].to(device=dev, dtype=torch.float32).contiguous()

        W["patch.w"] = mat("patch_embedding.weight").reshape(cfg.dim, -1).contiguous()
        W["patch.b"] = vec("patch_embedding.bias")
        for n in ("text_embedding.0", "text_embedding.2", "time_embedding.0", "time_embedding.2", "time_projection.1",
                  "img_emb.proj.1", "img_emb.proj.3", "head.head"):
            W[n + ".w"] = mat(n + ".weight")
            W[n + ".b"] = vec(n + ".bias")
        for n in ("img_emb.proj.0", "img_emb.proj.4"):
            W[n + ".w"] = vec(n + ".weight")
            W[n + ".b"] = vec(n + ".bias")
        W["head.modulation"] = vec("head.modulation").reshape(2, cfg.dim).contiguous()
        for i in range(cfg.num_layers):
            p = f"blocks.{i}."
            W[p + "qkv.w"] = torch.cat([mat(p + f"self_attn.{n}.weight") for n in "qkv"], dim=0).contiguous()
            W[p + "qkv.b"] = torch.cat([vec(p + f"self_attn.{n}.bias") for n in "qkv"], dim=0).contiguous()
            W[p + "self_attn.o.w"] = mat(p + "self_attn.o.weight")
            W[p + "self_attn.o.b"] = vec(p + "self_attn.o.bias")
            W[p + "self_attn.norm_q"] = vec(p + "self_attn.norm_q.weight")
            W[p + "self_attn.norm_k"] = vec(p + "self_attn.norm_k.weight")
            for n in ("q", "o"):
                W[p + f"cross_attn.{n}.w"] = mat(p + f"cross_attn.{n}.weight")
                W[p + f"cross_attn.{n}.b"] = vec(p + f"cross_attn.{n}.bias")
            W[p + "cross_attn.kv.w"] = torch.cat([mat(p + "cross_attn.k.weight"), mat(p + "cross_attn.v.weight")], 0).contiguous()
            W[p + "cross_attn.kv.b"] = torch.cat([vec(p + "cross_attn.k.bias"), vec(p + "cross_attn.v.bias")], 0).contiguous()
            W[p + "cross_attn.kv_img.w"] = torch.cat([mat(p + "cross_attn.k_img.weight"), mat(p + "cross_attn.v_img.weight")], 0).contiguous()
            W[p + "cross_attn.kv_img.b"] = torch.cat([vec(p + "cross_attn.k_img.bias"), vec(p + "cross_attn.v_img.bias")], 0).contiguous()
            for n in ("norm_q", "norm_k", "norm_k_img"):
                W[p + "cross_attn." + n] = vec(p + f"cross_attn.{n}.weight")
            W[p + "norm3.w"] = vec(p + "norm3.weight")
            W[p + "norm3.b"] = vec(p + "norm3.bias")
            for n in ("ffn.0", "ffn.2"):
                W[p + n + ".w"] = mat(p + n + ".weight")
                W[p + n + ".b"] = vec(p + n + ".bias")
            fp = ffn_padded_features(cfg.ffn_dim)  # zero rows so that FFN-up runs the 320-feature GEMM tile
            W[p + "ffn.0.w"], W[p + "ffn.0.b"] = _pad_rows(W[p + "ffn.0.w"], fp), _pad_rows(W[p + "ffn.0.b"], fp)
            W[p + "modulation"] = vec(p + "modulation").reshape(6, cfg.dim).contiguous()
        self.w = W
        return self

    def load_diffusers_state_dict(self, sd: Dict[str, torch.Tensor]):
        km = diffusers_key_map(self.cfg.num_layers)
        out = {}
        for k, v in sd.items():
            if k == "scale_shift_table":
                out["head.modulation"] = v
                continue
            if k.endswith(".scale_shift_table"):
                out[k.replace(".scale_shift_table", ".modulation")] = v
                continue
            base, _, leaf = k.rpartition(".")
            if base not in km:
                raise KeyError(f"unmapped diffusers parameter {k}")
            out[f"{km[base]}.{leaf}"] = v
        return self.load_state_dict(out)

    @classmethod
    def from_pretrained(cls, path: str, device="cuda:0", comm=None, subfolder: str = "transformer"):
        """The `transformer/` component of a local diffusers Wan2.1-I2V checkpoint directory (what
        `WanImageToVideoPipeline.from_pretrained(model_id, ...)` loads at INFER:191-197): `config.json` (diffusers field names:
        num_attention_heads x attention_head_dim, ffn_dim, num_layers, in_channels, out_channels, text_dim, freq_dim, image_dim,
        eps -- recalled from diffusers 0.35, not verifiable offline; missing fields keep the Wan2.1-I2V-14B values of
        wan/configs/wan_i2v_14B.py:27-36) + safetensors shards through checkpoint.load_dir and the diffusers key map."""
        import json as _json
        from . import checkpoint
        folder = os.path.join(path, subfolder) if subfolder and os.path.isdir(os.path.join(path, subfolder)) else path
        cfg = DiTConfig.wan_i2v_14b()
        cj = os.path.join(folder, "config.json")
        if os.path.exists(cj):
            with open(cj) as f:
                c = _json.load(f)
            heads = int(c.get("num_attention_heads", cfg.num_heads))
            cfg = DiTConfig(dim=heads * int(c.get("attention_head_dim", 128)), ffn_dim=int(c.get("ffn_dim", cfg.ffn_dim)), num_heads=heads,
                            num_layers=int(c.get("num_layers", cfg.num_layers)), in_dim=int(c.get("in_channels", cfg.in_dim)),
                            out_dim=int(c.get("out_channels", cfg.out_dim)), freq_dim=int(c.get("freq_dim", cfg.freq_dim)),
                            text_dim=int(c.get("text_dim", cfg.text_dim)), img_dim=int(c.get("image_dim") or cfg.img_dim),
                            patch_size=tuple(c.get("patch_size", cfg.patch_size)), eps=float(c.get("eps", cfg.eps)))
        return cls(cfg, device, comm=comm).load_diffusers_state_dict(checkpoint.load_dir(folder))

    def init_random(self, seed: int = 0):
        """Synthetic weights of the right shapes, generated directly on the device: there are no checkpoints
        offline.  Scaled so activations stay O(1) through 40 layers."""
        cfg, dev = self.cfg, self.device
        g = torch.Generator(device=dev).manual_seed(seed)
        W = {}

        def mat(n, k, std=None):
            std = std if std is not None else 1.0 / math.sqrt(k)
            return (torch.randn(n, k, generator=g, device=dev, dtype=torch.float32) * std).to(torch.bfloat16)

        def vec(n, std=0.02, base=0.0):
            return (torch.randn(n, generator=g, device=dev, dtype=torch.float32) * std + base).to(torch.bfloat16).float()

        d, f = cfg.dim, cfg.ffn_dim
        kp = cfg.in_dim * 4
        W["patch.w"], W["patch.b"] = mat(d, kp), vec(d)
        for n, (o, i) in {"text_embedding.0": (d, cfg.text_dim), "text_embedding.2": (d, d), "time_embedding.0": (d, cfg.freq_dim),
                          "time_embedding.2": (d, d), "time_projection.1": (6 * d, d), "img_emb.proj.1": (cfg.img_dim, cfg.img_dim),
                          "img_emb.proj.3": (d, cfg.img_dim), "head.head": (4 * cfg.out_dim, d)}.items():
            W[n + ".w"], W[n + ".b"] = mat(o, i), vec(o)
        W["img_emb.proj.0.w"], W["img_emb.proj.0.b"] = vec(cfg.img_dim, 0.05, 1.0), vec(cfg.img_dim)
        W["img_emb.proj.4.w"], W["img_emb.proj.4.b"] = vec(d, 0.05, 1.0), vec(d)
        W["head.modulation"] = vec(2 * d, 1.0 / math.sqrt(d)).reshape(2, d).contiguous()
        for i in range(cfg.num_layers):
            p = f"blocks.{i}."
            W[p + "qkv.w"], W[p + "qkv.b"] = mat(3 * d, d), vec(3 * d)
            W[p + "self_attn.o.w"], W[p + "self_attn.o.b"] = mat(d, d), vec(d)
            W[p + "self_attn.norm_q"], W[p + "self_attn.norm_k"] = vec(d, 0.05, 1.0), vec(d, 0.05, 1.0)
            for n in ("q", "o"):
                W[p + f"cross_attn.{n}.w"], W[p + f"cross_attn.{n}.b"] = mat(d, d), vec(d)
            W[p + "cross_attn.kv.w"], W[p + "cross_attn.kv.b"] = mat(2 * d, d), vec(2 * d)
            W[p + "cross_attn.kv_img.w"], W[p + "cross_attn.kv_img.b"] = mat(2 * d, d), vec(2 * d)
            for n in ("norm_q", "norm_k", "norm_k_img"):
                W[p + "cross_attn." + n] = vec(d, 0.05, 1.0)
            W[p + "norm3.w"], W[p + "norm3.b"] = vec(d, 0.05, 1.0), vec(d)
            fp = ffn_padded_features(f)
            W[p + "ffn.0.w"], W[p + "ffn.0.b"] = _pad_rows(mat(f, d), fp), _pad_rows(vec(f), fp)
            W[p + "ffn.2.w"], W[p + "ffn.2.b"] = mat(d, f), vec(d)
            W[p + "modulation"] = vec(6 * d, 1.0 / math.sqrt(d)).reshape(6, d).contiguous()
        self.w = W
        return self

    def local_tokens(self, L: int) -> int:
        """Query rows this rank processes (all of them on one GPU)."""
        if self.comm is None:
            return L
        from .parallel import shard_plan
        return shard_plan(L, self.comm.world).local_tokens(self.comm.rank)

    def param_bytes(self) -> int:
        pad = ffn_padded_features(self.cfg.ffn_dim) - self.cfg.ffn_dim  # zero rows of FFN-up (weight bf16 + bias f32) are layout, not parameters
        return sum(t.numel() * t.element_size() for t in self.w.values()) - self.cfg.num_layers * pad * (self.cfg.dim * 2 + 4)

    # ------------------------------------------------------------------------------------------------------------
    # workspaces (allocated once per token count; everything stays resident in HBM)
    # ------------------------------------------------------------------------------------------------------------
    def _buf(self, name, shape, dtype, zero=False):
        key = (name, tuple(shape), dtype)
        t = self._ws.get(key)
        if t is None:
            t = (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=self.device)
            self._ws[key] = t
        return t

    def _context_kv(self, text: torch.Tensor, img: torch.Tensor):
        """Per-layer cross-attention K / V cache for one (text, image) context: {layer: (k, vT)} in the fused [image | text] layout.  The
        K / V of the prompt context depend on the prompt only -- not on the latents, the timestep or the step -- so the 130 forwards of a
        video compute them once per prompt (the reference recomputes them in every forward, model.py:215-218; the values are the same
        kernels on the same inputs: bit-identical, tests/test_gpu_dit.py).  Keyed by the identity AND version of the embedding storages
        and by the weights version (`w` setter / weights_changed()), so an in-place edit of the embeddings, a weight reload or a declared
        in-place weight edit starts a fresh entry; at most 4 contexts are kept
        (positive / negative prompt of the last two videos; 17 MB per layer each).  On by default since round 5 (+0.8 % steps/s at the
        81-frame 480p configuration; bench.py states it in `config.ctx_cache`); WF_CTX_CACHE=0 recomputes per forward -- the reference
        arm of the bit-equality test."""
        if os.environ.get("WF_CTX_CACHE", "1") == "0":
            return None
        key = (text.data_ptr(), text._version, tuple(text.shape), img.data_ptr(), img._version, tuple(img.shape), self._wver, self.cfg.num_layers, id(self.comm))
        cache = self.__dict__.setdefault("_ctx_cache", {})
        hit = cache.get(key)
        if hit is None:
            while len(cache) >= 4:
                cache.pop(next(iter(cache)))
            # the tensors are held so that their storage (and hence data_ptr) cannot be recycled while the entry lives
            hit = cache[key] = {"_keep": (text, img)}
        return hit

    def _exchange(self, tag: str, H: int, shard_len: int, mode: str, chunks: int):
        """The K / V^T exchange buffers of the forward `tag` (parallel.KVExchange), allocated once per shape and mode."""
        from .parallel import KVExchange
        key = ("kvx" + tag, H, shard_len, mode, chunks, id(self.comm))
        ex = self._ws.get(key)
        if ex is None:
            ex = self._ws[key] = KVExchange(self.comm, H, shard_len, mode, chunks, self.device)
        return ex

    def _rope_tables(self, f, h, w):
        key = (f, h, w)
        if key not in self._rope:
            c, s = rope_tables(128, f, h, w)
            self._rope[key] = (c.to(self.device), s.to(self.device))
        return self._rope[key]

    # ------------------------------------------------------------------------------------------------------------
    # forward
    # ------------------------------------------------------------------------------------------------------------
    def _embed_condition(self, t_value: float, text: torch.Tensor, img: torch.Tensor, tag: str = ""):
        """model.py:546-563.  Returns e [1,dim] f32, e0 [6,dim] f32, ctx_text bf16 [512,dim], ctx_img bf16 [n_img,dim]."""
        cfg, W, dev = self.cfg, self.w, self.device
        bf, f32 = torch.bfloat16, torch.float32
        _buf = lambda name, shape, dtype, zero=False: self._buf(name + tag, shape, dtype, zero)  # noqa: E731
        sin = sinusoidal_embedding_1d(cfg.freq_dim, t_value).to(dev)
        t0 = _buf("t0", (1, cfg.dim), f32)
        gemm(ops.cast(sin, bf), W["time_embedding.0.w"], W["time_embedding.0.b"], t0, EPI_F32)
        t0a = self._act(t0, None, bf, 0)
        e = _buf("e", (1, cfg.dim), f32)
        gemm(t0a, W["time_embedding.2.w"], W["time_embedding.2.b"], e, EPI_F32)
        e0 = _buf("e0", (1, 6 * cfg.dim), f32)
        gemm(self._act(e, None, bf, 0), W["time_projection.1.w"], W["time_projection.1.b"], e0, EPI_F32)
        # text: zero-pad to text_len rows (model.py:554-559), Linear -> GELU(tanh) -> Linear
        n_txt = text.shape[0]
        tx = _buf("txt_in", (cfg.text_len, cfg.text_dim), bf, zero=True)
        tx[:n_txt].copy_(text)
        if n_txt < cfg.text_len:
            tx[n_txt:].zero_()
        th = _buf("txt_h", (cfg.text_len, cfg.dim), bf)
        gemm(tx, W["text_embedding.0.w"], W["text_embedding.0.b"], th, EPI_BF16_GELU)
        ctx_t = _buf("ctx_t", (cfg.text_len, cfg.dim), bf)
        gemm(th, W["text_embedding.2.w"], W["text_embedding.2.b"], ctx_t, EPI_BF16)
        # image: LayerNorm -> Linear -> GELU(erf) -> Linear -> LayerNorm (model.py:355-358)
        n_img = img.shape[0]
        i0 = _buf("img_ln0", (n_img, cfg.img_dim), bf)
        self._ln(ops.cast(img, f32), W["img_emb.proj.0.w"], W["img_emb.proj.0.b"], i0, 1e-5, plus_one=False)
        i1 = _buf("img_h", (n_img, cfg.img_dim), f32)
        gemm(i0, W["img_emb.proj.1.w"], W["img_emb.proj.1.b"], i1, EPI_F32)
        i2 = _buf("img_o", (n_img, cfg.dim), f32)
        gemm(self._act(i1, None, bf, 1), W["img_emb.proj.3.w"], W["img_emb.proj.3.b"], i2, EPI_F32)
        ctx_i = _buf("ctx_i", (n_img, cfg.dim), bf)
        self._ln(i2, W["img_emb.proj.4.w"], W["img_emb.proj.4.b"], ctx_i, 1e-5, plus_one=False)
        return e, e0.view(6, cfg.dim), ctx_t, ctx_i

    def _act(self, a, b, out_dtype, mode):
        out = torch.empty(a.shape, dtype=out_dtype, device=a.device)
        dt = {torch.float32: WF_F32, torch.bfloat16: WF_BF16}
        call("wf_act", a.data_ptr(), dt[a.dtype], b.data_ptr() if b is not None else None, dt[b.dtype] if b is not None else 0,
             out.data_ptr(), dt[out_dtype], mode, a.numel(), ops.stream())
        return out

    def _ln(self, x, mul, add, out, eps, plus_one):
        L, C = x.shape
        assert x.dtype == torch.float32 and x.is_contiguous() and out.is_contiguous()
        call("wf_ln_modulate", x.data_ptr(), mul.data_ptr() if mul is not None else None,
             add.data_ptr() if add is not None else None, out.data_ptr(), WF_BF16 if out.dtype == torch.bfloat16 else WF_F32,
             L, C, float(eps), 1 if plus_one else 0, ops.stream())
        return out

    def _heads(self, src, col0, weight, cos, sin, out, L, out_scale: float = 1.0, lout: Optional[int] = None,
               bound: Optional[torch.Tensor] = None):
        """RMSNorm(+RoPE) of columns [col0, col0+dim) of src [L, ld] -> out [H, Lout, 128]; out_scale: see wf_rmsnorm_heads.  lout: rows
        between two heads of the destination when `out` is a row range of a larger [H, lout, 128] buffer (the fused cross-attention keys).
        bound: f32 [H] that receives max over the rows of |out row|^2 per head, computed in the same pass (wf_rmsnorm_heads_bound) -- the
        norm bound of the attention kernel without wf_head_max_norm2's second pass over the tensor."""
        C = self.cfg.dim
        view = src[:, col0:col0 + C]
        if bound is not None:
            assert bound.dtype == torch.float32 and bound.numel() == self.cfg.num_heads and bound.is_contiguous()
            ws = self._buf("heads_bound_ws", (int(_ffi.lib().wf_rmsnorm_heads_bound_ws_floats(L, C)),), torch.float32)
            call("wf_rmsnorm_heads_bound", view.data_ptr(), src.stride(0), weight.data_ptr(), cos.data_ptr() if cos is not None else None,
                 sin.data_ptr() if sin is not None else None, out.data_ptr(), L, out.shape[1] if lout is None else lout, C, float(self.cfg.eps),
                 float(out_scale), ws.data_ptr(), bound.data_ptr(), ops.stream())
            return
        call("wf_rmsnorm_heads", view.data_ptr(), src.stride(0), weight.data_ptr(), cos.data_ptr() if cos is not None else None,
             sin.data_ptr() if sin is not None else None, out.data_ptr(), L, out.shape[1] if lout is None else lout, C, float(self.cfg.eps),
             float(out_scale), ops.stream())

    def _vt(self, src, col0, out, L, hstride: Optional[int] = None):
        """hstride: 64-key tiles between two heads of the destination when `out` is a tile range of a larger buffer."""
        view = src[:, col0:col0 + self.cfg.dim]
        if hstride is None:
            call("wf_v_transpose", view.data_ptr(), src.stride(0), out.data_ptr(), L, out.shape[1] * 64, self.cfg.num_heads, ops.stream())
        else:
            call("wf_v_transpose_seg", view.data_ptr(), src.stride(0), out.data_ptr(), L, out.shape[1] * 64, self.cfg.num_heads, hstride,
                 ops.stream())

    # How a forward WITHOUT a second CFG branch exchanges K / V^T (parallel.KVExchange): "chunked" (default, `exchange_chunks` all-gathers,
    # own shard first), "bcast" (per-source broadcasts, own shard first) or "gather" (one all-gather, one launch: bit-identical to one GPU
    # up to 4 ranks).  The lock-step CFG pair always uses "gather": the other branch's layer hides the exchange.  bench.py --gpus N times
    # the modes on the node at start-up and sets these (bench.calibrate_exchange).
    exchange_mode = "chunked"
    exchange_chunks = 2
    pair_lockstep = True      # forward_tokens_pair under sequence parallelism: two forwards one layer apart (False: one after the other)
    pair_share_layer0 = True  # forward_tokens_pair: the prompt-independent prefix (up to layer 0's self-attention block) computed once per pair
    attn_prescale = True      # softmax_scale * log2(e) folded into Q by its producer (k_attn_w4<4>); False: in-kernel scale (k_attn_w4<0>)
    attn_track_max = False    # True: withhold the norm bounds -> the kernel's max-tracking body (A/B and the tracked-body parity tests)

    def forward_tokens(self, x_in: torch.Tensor, t_value: float, text: torch.Tensor, img: torch.Tensor) -> torch.Tensor:
        """x_in [in_dim, T, h, w] bf16; text [<=512, text_dim]; img [n_img, img_dim] -> velocity [out_dim, T, h, w] f32."""
        out = [None]
        for _ in self._forward_steps(x_in, t_value, text, img, "", out, self.exchange_mode):
            pass
        return out[0]

    def forward_tokens_pair(self, x_in: torch.Tensor, t_value: float, text_a: torch.Tensor, text_b: torch.Tensor,
                            img: torch.Tensor, interleave: Optional[bool] = None):
        """The two forwards of one classifier-free-guidance evaluation (PIPE:593-610: same latents and timestep, positive then
        negative prompt).  Under sequence parallelism the two are advanced in lock-step, one layer apart: while branch A's K / V^T
        all-gather of layer i is in flight on the communication stream, branch B computes its layer i-1 attention / FFN, and vice
        versa, so every exchange has a whole layer of the other branch to hide under.  Each branch issues exactly the kernels of
        forward_tokens (exchange mode "gather") in the same order on its own buffers: results are bit-identical to two sequential calls."""
        if interleave is None:
            interleave = self.comm is not None and self.pair_lockstep
        # What the two branches compute IDENTICALLY is computed once (round 6): same latents and timestep, different prompt
        # (PIPE:593-610) -- patch embedding, time embedding / modulation and layer 0's whole self-attention block (LN-modulate, QKV,
        # RoPE / norm, attention, O projection: model.py:298-306) do not see the prompt; the first prompt-dependent operation is layer 0's
        # cross-attention.  Branch A leaves its residual stream after that block in a buffer, branch B starts from it: the same kernels on
        # the same inputs, so bit-identical to two independent forwards (tests/test_gpu_dit.py); ~0.8 % of a CFG evaluation.
        share = {} if self.pair_share_layer0 else None
        if not interleave:
            oa, ob = [None], [None]
            for _ in self._forward_steps(x_in, t_value, text_a, img, "", oa, self.exchange_mode, share, "produce"):
                pass
            for _ in self._forward_steps(x_in, t_value, text_b, img, "", ob, self.exchange_mode, share, "consume"):
                pass
            return oa[0], ob[0]
        oa, ob = [None], [None]
        ga = self._forward_steps(x_in, t_value, text_a, img, "", oa, "gather", share, "produce")
        gb = self._forward_steps(x_in, t_value, text_b, img, "#b", ob, "gather", share, "consume")
        live = [ga, gb]
        while live:
            for gen in list(live):
                try:
                    next(gen)
                except StopIteration:
                    live.remove(gen)
        return oa[0], ob[0]

    def _forward_steps(self, x_in, t_value, text, img, tag, result, mode="gather", share=None, role=None):
        """Generator over one forward: yields once per layer, right after that layer's K / V^T exchange has been launched (the
        point where another forward can usefully take over the compute stream).  `tag` separates the workspaces of concurrent
        forwards; the velocity lands in result[0].  share / role: the two forwards of a CFG pair (forward_tokens_pair) -- the "produce"
        forward copies its residual stream after layer 0's self-attention block into share["x"], the "consume" forward starts from it
        instead of computing the patch embedding and that block again."""
        cfg, W, dev = self.cfg, self.w, self.device
        _buf = lambda name, shape, dtype, zero=False: self._buf(name + tag, shape, dtype, zero)  # noqa: E731
        bf, f32 = torch.bfloat16, torch.float32
        Cin, T, Hh, Ww = x_in.shape
        assert Cin == cfg.in_dim
        f, h2, w2 = T, Hh // 2, Ww // 2
        Lfull = f * h2 * w2
        d, H = cfg.dim, cfg.num_heads
        scale = 1.0 / math.sqrt(128.0)
        # self-attention: softmax_scale * log2(e) is folded into Q by its producer (in front of the one bf16 rounding) and the attention
        # kernel is told so with softmax_scale = 0 (k_attn_w4<4>: score accumulators start from -m)
        prescale = bool(self.attn_prescale)
        q_scale, sa_scale = (scale * 1.4426950408889634, 0.0) if prescale else (1.0, scale)
        cos, sin = self._rope_tables(f, h2, w2)
        e, e0, ctx_t, ctx_i = self._embed_condition(t_value, text, img, tag)
        n_img = ctx_i.shape[0]
        comm = self.comm
        tok = _buf("tok", (Lfull, Cin * 4), bf)
        call("wf_patchify", x_in.data_ptr(), tok.data_ptr(), Cin, T, Hh, Ww, ops.stream())
        if comm is not None:
            # sequence parallelism: this rank owns a contiguous token shard; K / V^T are exchanged per layer (parallel.py)
            from .parallel import shard_plan
            plan = shard_plan(Lfull, comm.world)
            lo, hi = plan.bounds(comm.rank)
            L, Lp = hi - lo, plan.shard_len
            if L <= 0:
                raise ValueError(f"sequence-parallel plan leaves rank {comm.rank} of {comm.world} without tokens "
                                 f"({Lfull} tokens in shards of {Lp}); use fewer ranks for this size")
            tok, cos, sin = tok[lo:hi], cos[lo:hi], sin[lo:hi]
        else:
            plan = None
            L, Lp = Lfull, _pad64(Lfull)

        # patch embedding (model.py:534-537) as a GEMM -> fp32 residual stream
        x = _buf("x", (L, d), f32)
        consume = share is not None and role == "consume"
        if not consume:
            gemm(tok, W["patch.w"], W["patch.b"], x, EPI_F32)

        hbuf = _buf("h", (L, d), bf)
        qkv = _buf("qkv", (L, 3 * d), bf)
        qh = _buf("qh", (H, L, 128), bf)
        # per-head max |k|^2 / |q|^2 (out of the producers' own pass: wf_rmsnorm_heads_bound) let the kernel drop its running-max tracking
        # when no score can overflow; attn_track_max withholds them (A/B)
        use_bounds = prescale and not self.attn_track_max
        qm = _buf("qmax2", (H,), f32) if use_bounds else None
        ex = None
        if comm is not None:
            if not prescale or comm.world == 1:
                mode = "gather"   # part launches are built for the pre-scaled-Q form
            ex = self._exchange(tag, H, Lp, mode, int(self.exchange_chunks))
        else:
            kh = _buf("kh", (H, Lp, 128), bf, zero=True)
            vt = _buf("vt", (H, Lp // 64, 128, 64), bf)
            km = _buf("kmax2", (H,), f32) if use_bounds else None
        ao = _buf("ao", (L, d), bf)
        qc = _buf("qc", (L, d), bf)
        ffh = _buf("ffh", (L, ffn_padded_features(cfg.ffn_dim)), bf)
        Lt, Li = cfg.text_len, _pad64(n_img)
        # the two cross-attentions of a layer (image context, then text context, summed: model.py:220-227) are ONE launch over a concatenated
        # key / value buffer [image tiles | text tiles] (wf_attn_cross2_fwd; bit-identical to two wf_attn_fwd launches: tests/test_gpu_dit.py)
        Lc = Li + Lt
        kvt = _buf("kvt", (cfg.text_len, 2 * d), bf)
        kvi = _buf("kvi", (n_img, 2 * d), bf)
        kc = _buf("kc", (H, Lc, 128), bf, zero=True)
        vtc = _buf("vtc", (H, Lc // 64, 128, 64), bf)
        emod = _buf("emod", (6, d), f32)

        def context_operands(pl, kc_, vtc_):
            """The fused layout of one layer's prompt-context keys / values: rows [0, Li) image, [Li, Li + Lt) text."""
            gemm(ctx_i, W[pl + "cross_attn.kv_img.w"], W[pl + "cross_attn.kv_img.b"], kvi, EPI_BF16)
            self._heads(kvi, 0, W[pl + "cross_attn.norm_k_img"], None, None, kc_[:, :Li], n_img, lout=Lc)
            self._vt(kvi, d, vtc_[:, :Li // 64], n_img, hstride=Lc // 64)
            gemm(ctx_t, W[pl + "cross_attn.kv.w"], W[pl + "cross_attn.kv.b"], kvt, EPI_BF16)
            self._heads(kvt, 0, W[pl + "cross_attn.norm_k"], None, None, kc_[:, Li:], Lt, lout=Lc)
            self._vt(kvt, d, vtc_[:, Li // 64:], Lt, hstride=Lc // 64)

        ctx_kv = self._context_kv(text, img)
        # Sequence-parallel jobs: the K / V of the prompt context (text 512 rows, image 257) are the same on every rank and do not shrink with
        # the token shard -- 3 % of a rank's GPU time at 8 ranks when every rank computes all 40 layers' (measured).  Each
        # rank computes the layers i = rank (mod P) only, the finished kernel operands ([H, rows, 128] keys, blocked V^T) are all-gathered
        # once per forward on the communication stream; layer i then reads slot [i % P][i // P].  Same kernels per layer: bit-identical.
        # With the context cache (default) this happens on the first forward of a prompt only: the gathered buffers ARE the cache entry.
        ctx_shared = ctx_events = None
        share_ctx = comm is not None and comm.world > 1
        if share_ctx and ctx_kv is not None and "shared" in ctx_kv:
            (ctx_shared, ctx_events), share_ctx = ctx_kv["shared"], False   # (the events have long fired; waiting again costs nothing)

        def launch_context():
            """Called once, right after layer 0's K / V^T exchange has been launched: the context gathers queue BEHIND it on the communication
            stream (layer 0's self-attention needs its keys first, the context is not read before layer 0's cross-attention)."""
            P_, nl = comm.world, (cfg.num_layers + comm.world - 1) // comm.world
            mk = (lambda name, shape, zero=False: (torch.zeros if zero else torch.empty)(shape, dtype=bf, device=dev)) if ctx_kv is not None \
                else (lambda name, shape, zero=False: _buf(name, shape, bf, zero))   # cached entries own their buffers
            loc = [_buf("ckvf_loc0", (nl, H, Lc, 128), bf, zero=True), _buf("ckvf_loc1", (nl, H, Lc // 64, 128, 64), bf)]
            allb = [mk(f"ckvf_all{j}", (P_,) + tuple(t.shape)) for j, t in enumerate(loc)]
            for j in range(nl):
                li = comm.rank + P_ * j
                if li >= cfg.num_layers:
                    break
                context_operands(f"blocks.{li}.", loc[0][j], loc[1][j])
            shared = ((allb, P_), [comm.all_gather_async(a_, l_) for a_, l_ in zip(allb, loc)])
            if ctx_kv is not None:
                ctx_kv["shared"] = shared
            return shared

        for i in range(cfg.num_layers):
            p = f"blocks.{i}."
            # e = modulation + e0 (model.py:298)
            call("wf_act", W[p + "modulation"].data_ptr(), WF_F32, e0.data_ptr(), WF_F32, emod.data_ptr(), WF_F32, 2,
                 emod.numel(), ops.stream())
            # ---- self-attention (model.py:302-306) ----
            if i == 0 and consume:
                # the other branch of the CFG pair has computed this block on the same latents and timestep: take its residual stream
                if comm is not None:
                    if share_ctx:      # this branch's prompt context, gathered behind the producer's layer-0 exchange
                        ctx_shared, ctx_events = launch_context()
                    yield i            # (lock-step: the producer runs one layer ahead; its copy is queued before this one resumes)
                x.copy_(share["x"])
            elif comm is None:
                self._ln(x, emod[1], emod[0], hbuf, cfg.eps, plus_one=True)
                gemm(hbuf, W[p + "qkv.w"], W[p + "qkv.b"], qkv, EPI_BF16)
                self._heads(qkv, 0, W[p + "self_attn.norm_q"], cos, sin, qh, L, out_scale=q_scale, bound=qm)
                self._heads(qkv, d, W[p + "self_attn.norm_k"], cos, sin, kh, L, bound=km)
                self._vt(qkv, 2 * d, vt, L)
                attention(qh, kh, vt, ao, L, sa_scale, profile=True, kmax2=km, qmax2=qm)
            else:
                self._ln(x, emod[1], emod[0], hbuf, cfg.eps, plus_one=True)
                # K and V first: the producers write this rank's shard (and its norm bounds) straight into its slot of the exchange
                # buffers, chunk by chunk; the exchange runs on the communication stream under the Q projection
                gemm(hbuf, W[p + "qkv.w"][d:], W[p + "qkv.b"][d:], qkv[:, d:], EPI_BF16)
                for g in range(ex.G):
                    r0, r1 = ex.chunk_rows(g, L)
                    if r1 > r0:
                        self._heads(qkv[r0:r1], d, W[p + "self_attn.norm_k"], cos[r0:r1], sin[r0:r1], ex.own_k(g), r1 - r0,
                                    bound=ex.own_km(g) if use_bounds else None)
                        self._vt(qkv[r0:r1], 2 * d, ex.own_vt(g), r1 - r0)
                ex.launch()
                if i == 0 and share_ctx:
                    ctx_shared, ctx_events = launch_context()
                yield i
                gemm(hbuf, W[p + "qkv.w"][:d], W[p + "qkv.b"][:d], qkv[:, :d], EPI_BF16)
                self._heads(qkv, 0, W[p + "self_attn.norm_q"], cos, sin, qh, L, out_scale=q_scale, bound=qm)
                attention_exchange(qh, ex, ao, Lfull, sa_scale, qm, use_bounds=use_bounds, profile=True)
            if not (i == 0 and consume):
                gemm(ao, W[p + "self_attn.o.w"], W[p + "self_attn.o.b"], x, EPI_RESID, gate=emod[2])
            if i == 0 and share is not None and role == "produce":
                share["x"] = self._buf("x_pair_layer0", (L, d), f32)
                share["x"].copy_(x)
            # ---- cross-attention (model.py:310, 202-229) ----
            self._ln(x, W[p + "norm3.w"], W[p + "norm3.b"], hbuf, cfg.eps, plus_one=False)
            gemm(hbuf, W[p + "cross_attn.q.w"], W[p + "cross_attn.q.b"], qc, EPI_BF16)
            self._heads(qc, 0, W[p + "cross_attn.norm_q"], None, None, qh, L)
            # K / V of the text and image context depend on the prompt only, not on the latents or the timestep: computed on
            # the first forward with a given (text, image) pair and kept (17 MB per layer) -- 130 forwards per video reuse them
            kv = ctx_kv.get(i) if ctx_kv is not None else None
            if ctx_shared is not None:
                if ctx_events is not None:  # first use: the gathers were launched before layer 0
                    for ev in ctx_events:
                        if ev is not None:
                            torch.cuda.current_stream().wait_event(ev)
                    ctx_events = None
                allb, P_ = ctx_shared
                kc, vtc = (a_[i % P_, i // P_] for a_ in allb)
            elif kv is None:
                if ctx_kv is not None:  # own buffers per layer
                    kc, vtc = torch.zeros((H, Lc, 128), dtype=bf, device=dev), torch.empty((H, Lc // 64, 128, 64), dtype=bf, device=dev)
                context_operands(p, kc, vtc)
                if ctx_kv is not None:
                    ctx_kv[i] = (kc, vtc)
            else:
                kc, vtc = kv
            cross_attention2(qh, kc, vtc, ao, Li, n_img, Lt, scale)
            gemm(ao, W[p + "cross_attn.o.w"], W[p + "cross_attn.o.b"], x, EPI_RESID, gate=None)
            # ---- FFN (model.py:311-313) ----
            self._ln(x, emod[4], emod[3], hbuf, cfg.eps, plus_one=True)
            gemm(hbuf, W[p + "ffn.0.w"], W[p + "ffn.0.b"], ffh, EPI_BF16_GELU)
            gemm(ffh[:, :cfg.ffn_dim], W[p + "ffn.2.w"], W[p + "ffn.2.b"], x, EPI_RESID, gate=emod[5])

        # ---- head (model.py:337-347) + unpatchify (:584-607) ----
        hm = _buf("hm", (2, d), f32)
        # head modulation (model.py:345): hm[r] = modulation[r] + e
        for r in range(2):
            call("wf_act", W["head.modulation"][r].data_ptr(), WF_F32, e.data_ptr(), WF_F32, hm[r].data_ptr(), WF_F32, 2, d,
                 ops.stream())
        self._ln(x, hm[1], hm[0], hbuf, cfg.eps, plus_one=True)
        y = _buf("y", (L, 4 * cfg.out_dim), f32)
        gemm(hbuf, W["head.head.w"], W["head.head.b"], y, EPI_F32)
        if comm is not None:
            from .parallel import gather_rows
            y = gather_rows(comm, y, plan).contiguous()
        out = torch.empty((cfg.out_dim, T, Hh, Ww), dtype=f32, device=dev)
        call("wf_unpatchify", y.data_ptr(), out.data_ptr(), cfg.out_dim, T, Hh, Ww, ops.stream())
        result[0] = out

    def __call__(self, hidden_states: torch.Tensor, timestep, encoder_hidden_states: torch.Tensor,
                 encoder_hidden_states_image: Optional[torch.Tensor] = None, attention_kwargs=None, return_dict: bool = False):
        if hidden_states.shape[0] != 1:
            raise NotImplementedError("batch size 1 (the reference path); run CFG branches as separate calls")
        t_value = float(torch.as_tensor(timestep).reshape(-1)[0].item())
        x = hidden_states[0]
        if x.dtype != torch.bfloat16:
            x = ops.cast(x.contiguous(), torch.bfloat16)
        v = self.forward_tokens(x.contiguous(), t_value, encoder_hidden_states[0].to(torch.bfloat16).contiguous(),
                                encoder_hidden_states_image[0].to(torch.bfloat16).contiguous())
        out = ops.cast(v, self.dtype).unsqueeze(0)
        if return_dict:
            return SimpleNamespace(sample=out)
        return (out,)

    def forward_cfg_pair(self, hidden_states: torch.Tensor, timestep, encoder_hidden_states: torch.Tensor,
                         negative_encoder_hidden_states: torch.Tensor, encoder_hidden_states_image: torch.Tensor):
        """Both transformer calls of PIPE:593-610 (positive, then negative prompt) -> (noise_pred, noise_uncond), each what
        __call__(...)[0] returns.  See forward_tokens_pair."""
        if hidden_states.shape[0] != 1:
            raise NotImplementedError("batch size 1 (the reference path)")
        t_value = float(torch.as_tensor(timestep).reshape(-1)[0].item())
        x = hidden_states[0]
        if x.dtype != torch.bfloat16:
            x = ops.cast(x.contiguous(), torch.bfloat16)
        img = encoder_hidden_states_image[0].to(torch.bfloat16).contiguous()
        va, vb = self.forward_tokens_pair(x.contiguous(), t_value, encoder_hidden_states[0].to(torch.bfloat16).contiguous(),
                                          negative_encoder_hidden_states[0].to(torch.bfloat16).contiguous(), img)
        return ops.cast(va, self.dtype).unsqueeze(0), ops.cast(vb, self.dtype).unsqueeze(0)
