"""Block-sparse attention of the LongCat-Video 720p refine pass, HIP-backed.

Mirror of longcat_video/block_sparse_attention/bsa_interface.py (BSA): `flash_attn_bsa_3d` :612-659 = 3D-block token permute (:600-610)
-> mean-pool gating (:169-179) -> block scores (:181-185) -> top-k block selection (:211-224) -> sparse attention (:538-560, the Triton
kernel of flash_attn_bsa_varlen_mask.py:174-285) -> inverse permute.  Token-sized work runs in libwf_hip.so (`wf_lc_mean_pool_blocks`,
`wf_gemm_bf16_batched` for the block scores, `wf_attn_bsa_fwd`, the permutes ride on `wf_lc_norm_heads` / `wf_gather_rows_bf16`); the
top-k selection and the index bookkeeping that turns it into per-workgroup block lists are one kernel too (`wf_bsa_topk_lists`, round
2); the cdf-threshold rule (variable-length selections) keeps its small integer tensor ops in torch on the device (no host sync).
"""
from __future__ import annotations

from typing import Dict, Sequence, Tuple

import torch

from . import ops
from ._ffi import call
from .dit import EPI_BF16

BLOCK = 128  # default block; 64 (chunk 4 x 4 x 4) is the other size the kernels take
_PERM: Dict[tuple, Tuple[torch.Tensor, torch.Tensor]] = {}


def block_permutation(T: int, H: int, W: int, chunk: Sequence[int], device) -> Tuple[torch.Tensor, torch.Tensor]:
    """BSA:600-604.  Returns (perm, pos) int32 on the device: x_blocks[i] = x[perm[i]], pos[token] = its row in block order."""
    t, h, w = chunk
    if T % t or H % h or W % w:
        raise ValueError(f"latent grid {(T, H, W)} is not a whole number of {tuple(chunk)} blocks")
    if t * h * w not in (64, 128):
        raise NotImplementedError("the sparse attention kernel takes 128- or 64-token blocks (e.g. 4 x 4 x 8, 4 x 4 x 4)")
    key = (T, H, W, t, h, w, str(device))
    if key not in _PERM:
        idx = torch.arange(T * H * W).view(T // t, t, H // h, h, W // w, w).permute(0, 2, 4, 1, 3, 5).reshape(-1)
        pos = torch.empty_like(idx)
        pos[idx] = torch.arange(idx.numel())
        _PERM[key] = (idx.to(device=device, dtype=torch.int32), pos.to(device=device, dtype=torch.int32))
    return _PERM[key]


def mean_pool(x: torch.Tensor, block: int = BLOCK) -> torch.Tensor:
    """BSA:169-179: [heads, L, 128] bf16 -> [heads, L / block, 128] bf16."""
    Hh, L, D = x.shape
    assert D == 128 and x.dtype == torch.bfloat16 and x.is_contiguous()
    out = torch.empty((Hh, L // block, D), dtype=torch.bfloat16, device=x.device)
    call("wf_lc_mean_pool_blocks", x.data_ptr(), out.data_ptr(), Hh, L, block, ops.stream())
    return out


def block_scores(q_cmp: torch.Tensor, k_cmp: torch.Tensor) -> torch.Tensor:
    """BSA:181-185: per head q_cmp k_cmp^T in bf16 -> view [heads, n_q, n_k] (of a buffer padded to a multiple of 8 columns)."""
    Hh, nq, D = q_cmp.shape
    nk = k_cmp.shape[1]
    nkp = (nk + 7) // 8 * 8
    kp = k_cmp
    if nkp != nk:
        kp = torch.zeros((Hh, nkp, D), dtype=k_cmp.dtype, device=k_cmp.device)
        kp[:, :nk] = k_cmp
    sc = torch.empty((Hh, nq, nkp), dtype=torch.bfloat16, device=q_cmp.device)
    # one launch for all heads (round 1 issued one small GEMM per head: ~130 launches per layer)
    call("wf_gemm_bf16_batched", q_cmp.data_ptr(), kp.data_ptr(), sc.data_ptr(), Hh, nq, nkp, D, D, D, nkp, nq * D, nkp * D, nq * nkp,
         EPI_BF16, ops.stream())
    return sc[:, :, :nk]


def select_topk(scores: torch.Tensor, sparsity: float) -> torch.Tensor:
    """BSA:211-224: indices of the int((1 - sparsity) * n_k) best key blocks per query block."""
    n = int((1 - sparsity) * scores.shape[-1])
    if n < 1:
        raise ValueError(f"sparsity {sparsity} leaves no key block of {scores.shape[-1]}")
    return torch.topk(scores, n)[1]


def select_cdf(scores: torch.Tensor, cdf_threshold: float, sparsity=None):
    """(Torch form, kept for tests and for WF_BSA_TORCH_SELECT=1; the product path is cdf_lists / wf_bsa_cdf_lists.)  BSA:226-263: softmax(score / sqrt(128)) block weights, descending; as many blocks as the cumulative weight needs to pass
    cdf_threshold, at least the top-k count when `sparsity` is given.  Returns (sorted indices [heads, n_q, n_k], counts [heads, n_q])."""
    w = torch.softmax(scores * (1 / 128 ** 0.5), dim=-1)
    ws = torch.sort(w, dim=-1, descending=True)
    # eager CPU torch (the parity target) keeps a cumsum's running sum in fp32 and rounds every OUTPUT to the tensor's dtype; a GPU scan
    # may round the running sum itself -- spelled out so that bf16 scores give the reference's counts on either device
    cdf = torch.cumsum(ws.values.float(), dim=-1).to(ws.values.dtype).float()
    thr = torch.full(cdf.shape[:-1] + (1,), cdf_threshold, dtype=torch.float32, device=cdf.device)
    num = torch.searchsorted(cdf, thr, right=True).squeeze(-1)
    if sparsity is not None:
        num = num.clamp_min(int((1 - sparsity) * scores.shape[-1]))
    return ws.indices, num.clamp_max(scores.shape[-1])


def group_lists(block_indices: torch.Tensor, n_k: int, block_lens: torch.Tensor = None, block: int = BLOCK, blocks_per_segment: int = None):
    """[heads, n_q, n_sel] selected key blocks per query block -> the per-workgroup lists `wf_attn_bsa_fwd` walks: one list per group of
    g = 256 / block consecutive query blocks (the 256 query rows of a workgroup) holding the union of the group's blocks in ascending
    order, entry = physical_block * 2^g + sum_i 2^i * (selected by the i-th query block), where physical_block is the block's position
    in the K / V^T buffers ([P][heads][S][128] with S = blocks_per_segment * block keys per rank shard; one segment = the whole sequence
    on one GPU): (b // bps) * heads * bps + head * bps + b % bps -- head and segment are folded in here so that the kernel does no index
    arithmetic.  Returns (lists int32 [heads, n_groups, max_entries], counts int32 [heads, n_groups], max_entries); max_entries =
    min(g * n_sel, n_k) is a shape-only bound: no host sync."""
    Hh, nq, nsel = block_indices.shape
    gs = 256 // block
    allow = torch.zeros((Hh, (nq + gs - 1) // gs * gs, n_k), dtype=torch.bool, device=block_indices.device)
    if block_lens is None:
        allow[:, :nq].scatter_(2, block_indices.long(), True)
    else:  # variable-length selections (cdf threshold): the first block_lens entries of each row
        use = torch.arange(nsel, device=block_indices.device).view(1, 1, -1) < block_lens.unsqueeze(-1)
        allow[:, :nq].scatter_(2, block_indices.long(), use)
    parts = [allow[:, i::gs] for i in range(gs)]
    union = parts[0]
    for pt in parts[1:]:
        union = union | pt
    counts = union.sum(dim=-1).to(torch.int32)
    order = torch.sort((~union).to(torch.uint8), dim=-1, stable=True)[1]  # selected blocks first, ascending
    max_entries = min(gs * nsel, n_k)  # (variable-length lists come with nsel = n_k: the bound is n_k)
    order = order[..., :max_entries]
    bps = blocks_per_segment or n_k
    head = torch.arange(Hh, device=order.device).view(Hh, 1, 1)
    phys = (order // bps) * (Hh * bps) + head * bps + order % bps
    entries = phys * (1 << gs)
    for i, pt in enumerate(parts):
        entries = entries + (pt.gather(2, order).long() << i)
    return entries.to(torch.int32).contiguous(), counts.contiguous(), max_entries


TOPK_MAX_BLOCKS = 2048  # wf_bsa_topk_lists keeps a row of scores in LDS


class SelectionMask:
    """The selection wf_bsa_topk_lists made, as it left the kernel: bit b of mask[head, query block, b // 32] = key block b selected.
    `.cpu()` gives what `torch.topk(...)[1]` would have given up to order: int64 [heads, n_q, n_sel], ascending (tests, oracle hand-over)."""

    def __init__(self, mask: torch.Tensor, n_k: int, n_sel: int):
        self.mask, self.n_k, self.n_sel = mask, n_k, n_sel

    def indices(self) -> torch.Tensor:
        bits = (self.mask.unsqueeze(-1) >> torch.arange(32, device=self.mask.device, dtype=torch.int32)) & 1
        sel = bits.flatten(2)[:, :, :self.n_k].bool()
        assert bool((sel.sum(-1) == self.n_sel).all())
        return torch.nonzero(sel)[:, 2].view(sel.shape[0], sel.shape[1], self.n_sel)

    def cpu(self) -> torch.Tensor:
        return self.indices().cpu()


def topk_lists(scores: torch.Tensor, sparsity: float, block: int = BLOCK, blocks_per_segment: int = None):
    """BSA:211-224 + group_lists in ONE launch (wf_bsa_topk_lists): scores [heads, n_q, n_k] bf16 (a view of the padded gating buffer) ->
    (lists, counts, max_entries, SelectionMask).  Same lists as group_lists(select_topk(scores, sparsity)) when no two scores tie at the
    n_sel-th place (then: ascending block index here)."""
    Hh, nq, nk = scores.shape
    n = int((1 - sparsity) * nk)
    if n < 1:
        raise ValueError(f"sparsity {sparsity} leaves no key block of {nk}")
    assert scores.dtype == torch.bfloat16 and scores.stride(2) == 1 and scores.stride(0) == nq * scores.stride(1)
    gs = 256 // block
    ng = (nq + gs - 1) // gs
    mx = min(gs * n, nk)
    lists = torch.empty((Hh, ng, mx), dtype=torch.int32, device=scores.device)
    counts = torch.empty((Hh, ng), dtype=torch.int32, device=scores.device)
    mask = torch.empty((Hh, nq, (nk + 31) // 32), dtype=torch.int32, device=scores.device)
    call("wf_bsa_topk_lists", scores.data_ptr(), scores.stride(1), Hh, nq, nk, n, block, blocks_per_segment or nk, lists.data_ptr(),
         counts.data_ptr(), mx, mask.data_ptr(), ops.stream())
    return lists, counts, mx, SelectionMask(mask, nk, n)


class SelectionMaskVar:
    """The selection wf_bsa_cdf_lists made: the mask of SelectionMask plus the per-row counts.  `.cpu()` gives (indices int64 [heads, n_q,
    max count] -- each row's selected blocks ascending, padded with 0 --, counts [heads, n_q]): the (idx, lens) pair select_cdf returns,
    up to order."""

    def __init__(self, mask: torch.Tensor, n_k: int, counts: torch.Tensor):
        self.mask, self.n_k, self.counts = mask, n_k, counts

    def cpu(self):
        bits = (self.mask.unsqueeze(-1) >> torch.arange(32, device=self.mask.device, dtype=torch.int32)) & 1
        sel = bits.flatten(2)[:, :, :self.n_k].bool().cpu()
        lens = self.counts.cpu().long()
        assert bool((sel.sum(-1) == lens).all())
        width = max(int(lens.max()), 1)
        order = torch.sort((~sel).to(torch.uint8), dim=-1, stable=True)[1][..., :width]  # host side, tests only: selected first, ascending
        keep = torch.arange(width).view(1, 1, -1) < lens.unsqueeze(-1)
        return torch.where(keep, order, torch.zeros_like(order)), lens


def cdf_lists(scores: torch.Tensor, cdf_threshold: float, sparsity=None, block: int = BLOCK, blocks_per_segment: int = None):
    """BSA:226-263 + group_lists in two launches of one entry point (wf_bsa_cdf_lists), no torch ops: scores [heads, n_q, n_k] bf16 ->
    (lists, counts, max_entries, SelectionMaskVar)."""
    Hh, nq, nk = scores.shape
    n_min = 0 if sparsity is None else int((1 - sparsity) * nk)
    assert scores.dtype == torch.bfloat16 and scores.stride(2) == 1 and scores.stride(0) == nq * scores.stride(1)
    gs = 256 // block
    ng = (nq + gs - 1) // gs
    lists = torch.empty((Hh, ng, nk), dtype=torch.int32, device=scores.device)
    counts = torch.empty((Hh, ng), dtype=torch.int32, device=scores.device)
    rows = torch.empty((Hh, nq), dtype=torch.int32, device=scores.device)
    mask = torch.empty((Hh, nq, (nk + 31) // 32), dtype=torch.int32, device=scores.device)
    call("wf_bsa_cdf_lists", scores.data_ptr(), scores.stride(1), Hh, nq, nk, float(cdf_threshold), n_min, block, blocks_per_segment or nk,
         lists.data_ptr(), counts.data_ptr(), nk, mask.data_ptr(), rows.data_ptr(), ops.stream())
    return lists, counts, nk, SelectionMaskVar(mask, nk, rows)


def sparse_attention_cdf(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, out: torch.Tensor, scores: torch.Tensor, cdf_threshold: float,
                         sparsity, scale: float, block: int = BLOCK) -> SelectionMaskVar:
    """sparse_attention with the cdf selection of `scores` made on the way (three launches, no torch ops)."""
    Hh, Lq, _ = q.shape
    if k.dim() == 4:
        Lkp, seg = k.shape[0] * k.shape[2], k.shape[2]
    else:
        Lkp = seg = k.shape[1]
    lists, counts, mx, sel = cdf_lists(scores, cdf_threshold, sparsity, block, seg // block)
    call("wf_attn_bsa_fwd", q.data_ptr(), k.data_ptr(), vt.data_ptr(), out.data_ptr(), Hh, Lq, Lkp, seg, out.stride(0), float(scale),
         lists.data_ptr(), counts.data_ptr(), mx, block, ops.stream())
    return sel


def sparse_attention_topk(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, out: torch.Tensor, scores: torch.Tensor, sparsity: float,
                          scale: float, block: int = BLOCK) -> SelectionMask:
    """sparse_attention with the top-k selection of `scores` [heads, Lq / block, n_k_blocks] made on the way (two launches, no torch ops)."""
    Hh, Lq, _ = q.shape
    if k.dim() == 4:
        Lkp, seg = k.shape[0] * k.shape[2], k.shape[2]
    else:
        Lkp = seg = k.shape[1]
    lists, counts, mx, sel = topk_lists(scores, sparsity, block, seg // block)
    call("wf_attn_bsa_fwd", q.data_ptr(), k.data_ptr(), vt.data_ptr(), out.data_ptr(), Hh, Lq, Lkp, seg, out.stride(0), float(scale),
         lists.data_ptr(), counts.data_ptr(), mx, block, ops.stream())
    return sel


def sparse_attention(q: torch.Tensor, k: torch.Tensor, vt: torch.Tensor, out: torch.Tensor, block_indices: torch.Tensor, scale: float,
                     n_k_blocks: int, block_lens: torch.Tensor = None, block: int = BLOCK):
    """q [heads, Lq, 128], k [heads, Lkp, 128] and vt [heads, Lkp/64, 128, 64] in block order -- or their all-gathered per-rank shards
    k [P, heads, S, 128], vt [P, heads, S/64, 128, 64] (keys in shard-major order); block_indices [heads, Lq/block, n_sel] over the
    first n_k_blocks key blocks -> out [Lq, ld] bf16 (block order)."""
    Hh, Lq, _ = q.shape
    if k.dim() == 4:
        Lkp, seg = k.shape[0] * k.shape[2], k.shape[2]
    else:
        Lkp = seg = k.shape[1]
    lists, counts, mx = group_lists(block_indices, n_k_blocks, block_lens, block, seg // block)
    call("wf_attn_bsa_fwd", q.data_ptr(), k.data_ptr(), vt.data_ptr(), out.data_ptr(), Hh, Lq, Lkp, seg, out.stride(0), float(scale),
         lists.data_ptr(), counts.data_ptr(), mx, block, ops.stream())
    return out
