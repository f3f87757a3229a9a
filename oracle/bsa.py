"""Oracle: LongCat-Video block-sparse attention (the 720p refine pass) in plain torch (CPU).  TEST INFRASTRUCTURE ONLY.

Restates /root/reference/longcat_for_worldforge/longcat_video/block_sparse_attention/bsa_interface.py (BSA):
  rearrange_THW_to_3d_block / rearrange_3d_block_to_THW :600-610, mean_pooling_compression :169-179, cal_score :181-185,
  get_select_indices_topk_from_score :211-224, _attention_bsa.forward :538-560, flash_attn_bsa_3d :612-659.
The permutes, the pooling and the top-k selection are pinned against those functions imported from the reference
(tests/golden/g14_bsa.npz, tools/make_goldens.py bsa).  The sparse attention itself is a Triton kernel in the reference
(flash_attn_bsa_varlen_mask.py:174-285): it is restated as what it computes -- for every query block, softmax(q k^T * sm_scale) over the
keys of the selected key blocks only, p cast to the value dtype before P V, row sums from the uncast p, 0 for an empty selection -- and
PINNED (round 2) against that kernel itself: Triton's interpreter (TRITON_INTERPRET=1; triton-rocm 3.6 is in the image) executes the
reference's unmodified @triton.jit functions on CPU tensors, launched by the reference's own attn_fwd_bsa_varlen_triton / flash_attn_bsa /
flash_attn_bsa_3d -- tests/golden/g18_bsa_triton.npz (tools/make_goldens.py bsa_triton; fp32 and fp16 tensors, 128- and 64-token blocks,
variable-length and empty lists, the 3D-block interface).
"""
from __future__ import annotations

from typing import Sequence, Tuple

import torch


def block_permutation(T: int, H: int, W: int, t: int, h: int, w: int) -> torch.Tensor:
    """BSA:600-604: index tensor `perm` with x_blocks = x[perm]: tokens of each (t x h x w) brick made contiguous, bricks in
    (Nt, Nh, Nw) order, tokens inside a brick in (t, h, w) order."""
    assert T % t == 0 and H % h == 0 and W % w == 0
    idx = torch.arange(T * H * W).view(T // t, t, H // h, h, W // w, w)
    return idx.permute(0, 2, 4, 1, 3, 5).reshape(-1)


def mean_pool(x: torch.Tensor, block: int) -> torch.Tensor:
    """BSA:169-179: [heads, S, D] -> [heads, S / block, D] in x's dtype (S is a multiple of the block on this path)."""
    Hh, S, D = x.shape
    assert S % block == 0
    return x.view(Hh, S // block, block, D).mean(dim=2)


def select_topk(q_cmp: torch.Tensor, k_cmp: torch.Tensor, sparsity: float) -> torch.Tensor:
    """BSA:181-185 + 211-224: score = q_cmp k_cmp^T (input dtype); the int((1 - sparsity) * n_k) best key blocks per query block."""
    score = torch.matmul(q_cmp, k_cmp.transpose(-1, -2))
    n = int((1 - sparsity) * score.shape[-1])
    return torch.topk(score, n)[1]


def select_cdf(q_cmp: torch.Tensor, k_cmp: torch.Tensor, cdf_threshold: float, sparsity=None):
    """BSA:226-263 (get_select_indices_cdf / _cdf_topk): block weights = softmax(score / sqrt(D)); blocks in descending weight order,
    as many as it takes for the cumulative weight to pass cdf_threshold (searchsorted right), at least the top-k count when sparsity is
    given too.  Returns (sorted block indices [heads, n_q, n_k], number selected [heads, n_q])."""
    score = torch.matmul(q_cmp, k_cmp.transpose(-1, -2))
    w = torch.softmax(score * (1 / q_cmp.shape[-1] ** 0.5), dim=-1)
    ws = torch.sort(w, dim=-1, descending=True)
    cdf = torch.cumsum(ws.values, dim=-1)
    thr = torch.full(cdf.shape[:-1] + (1,), cdf_threshold, dtype=cdf.dtype)
    num = torch.searchsorted(cdf, thr, right=True).squeeze(-1)
    if sparsity is not None:
        num = num.clamp_min(int((1 - sparsity) * score.shape[-1]))
    return ws.indices, num


def cdf_counts_bf16(score: torch.Tensor, cdf_threshold: float, sparsity=None) -> torch.Tensor:
    """BSA:234-243 / 253-266 as eager torch evaluates it on a BF16 score tensor (the reference's model dtype), every rounding point spelled
    out: x = bf16(score * sm_scale); w = bf16(softmax computed in fp32); weights sorted descending; cdf_k = bf16(sequential fp32 running
    sum); count = #{float(cdf_k) <= float32(threshold)} (`searchsorted(right=True)` on a non-decreasing sequence), at least the top-k
    count when sparsity is given.  Pinned bit-exactly against the reference's own function on bf16 scores (g14c; tests/test_oracle_bsa.py)."""
    assert score.dtype == torch.bfloat16
    xf = (score.float() * (1 / 128 ** 0.5)).bfloat16().float()
    e = torch.exp(xf - xf.max(-1, keepdim=True).values)
    w = (e / e.sum(-1, keepdim=True)).bfloat16()
    ws = torch.sort(w, dim=-1, descending=True).values.float()
    acc = torch.zeros(ws.shape[:-1])
    num = torch.zeros(ws.shape[:-1], dtype=torch.int64)
    thr = torch.tensor(cdf_threshold, dtype=torch.float32)
    for k in range(ws.shape[-1]):
        acc = acc + ws[..., k]
        num += (acc.bfloat16().float() <= thr)
    if sparsity is not None:
        num = num.clamp_min(int((1 - sparsity) * score.shape[-1]))
    return num


def sparse_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, block_indices: torch.Tensor, block_q: int, block_k: int,
                     scale: float, block_lens: torch.Tensor = None, p_dtype=None) -> torch.Tensor:
    """flash_attn_bsa_varlen_mask.py:236-285 as a masked dense softmax: q [heads, Sq, D], k / v [heads, Sk, D] (block order),
    block_indices [heads, Sq / block_q, n_sel] -> [heads, Sq, D] in q's dtype.  p_dtype: the kernel's `p.to(v.dtype)` (:259) for half
    tensors -- the un-normalised probabilities are rounded to it before P V while the row sums keep the fp32 p (:255)."""
    Hh, Sq, D = q.shape
    Sk = k.shape[1]
    nq, nk = Sq // block_q, Sk // block_k
    allow = torch.zeros((Hh, nq, nk), dtype=torch.bool)
    if block_lens is None:
        allow.scatter_(2, block_indices.long(), True)
    else:  # variable-length lists: the first block_lens entries of each (sorted) index row
        use = torch.arange(block_indices.shape[-1]).view(1, 1, -1) < block_lens.unsqueeze(-1)
        allow.scatter_(2, block_indices.long(), use)
    mask = allow.repeat_interleave(block_q, dim=1).repeat_interleave(block_k, dim=2)
    s = torch.einsum("hqd,hkd->hqk", q.float(), k.float()) * scale
    s = s.masked_fill(~mask, float("-inf"))
    if p_dtype is None:
        p = torch.softmax(s, dim=-1)
        p = torch.where(mask.any(dim=-1, keepdim=True), p, torch.zeros_like(p))  # empty selection: acc / l = 0 / 1 in the reference's kernel
        return torch.einsum("hqk,hkd->hqd", p, v.float()).to(q.dtype)
    m = s.max(dim=-1, keepdim=True).values
    m = torch.where(torch.isfinite(m), m, torch.zeros_like(m))
    pu = torch.exp(s - m)                      # un-normalised, <= 1 (the kernel's running max makes its p <= 1 too)
    l = pu.sum(dim=-1, keepdim=True)
    acc = torch.einsum("hqk,hkd->hqd", pu.to(p_dtype).float(), v.float())
    return (acc / torch.where(l > 0, l, torch.ones_like(l))).to(q.dtype)


def flash_attn_bsa_3d(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, shape_q: Tuple[int, int, int], shape_k: Tuple[int, int, int],
                      sparsity: float = 0.875, chunk_q: Sequence[int] = (4, 4, 8), chunk_k: Sequence[int] = (4, 4, 8),
                      return_indices: bool = False, gate_dtype=None, block_indices=None):
    """BSA:612-659 for one sample: q [heads, Sq, D], k / v [heads, Sk, D] in (T, H, W) token order -> [heads, Sq, D]."""
    pq, pk = block_permutation(*shape_q, *chunk_q), block_permutation(*shape_k, *chunk_k)
    bq, bk = chunk_q[0] * chunk_q[1] * chunk_q[2], chunk_k[0] * chunk_k[1] * chunk_k[2]
    qb, kb, vb = q[:, pq], k[:, pk], v[:, pk]
    if block_indices is not None:
        idx = block_indices
    elif gate_dtype is not None:  # the reference's q / k are bf16 on the GPU: pooled means and block scores are rounded to it
        idx = select_topk(mean_pool(qb.to(gate_dtype), bq), mean_pool(kb.to(gate_dtype), bk), sparsity)
    else:
        idx = select_topk(mean_pool(qb, bq), mean_pool(kb, bk), sparsity)
    ob = sparse_attention(qb, kb, vb, idx, bq, bk, q.shape[-1] ** -0.5)
    out = torch.empty_like(ob)
    out[:, pq] = ob  # BSA:606-610
    return (out, idx) if return_indices else out
