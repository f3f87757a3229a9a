"""Stand-in for the third-party `flash-attn` package (absent from this image) with its published semantics, in plain fp32 torch:
exact softmax attention.  Used only by tools/make_goldens.py so that the unmodified reference modules that call flash-attn can run
on the CPU.  Not reference code and not part of the product."""
import torch


def _sdpa(q, k, v, scale):
    # q [Sq, H, D], k / v [Sk, H, D]
    s = torch.einsum("qhd,khd->hqk", q.float(), k.float()) * scale
    return torch.einsum("hqk,khd->qhd", torch.softmax(s, dim=-1), v.float()).to(q.dtype)


def flash_attn_func(q, k, v, dropout_p=0.0, softmax_scale=None, causal=False, **kw):
    """q, k, v [B, S, H, D] -> [B, S, H, D]."""
    assert dropout_p == 0.0 and not causal
    scale = softmax_scale if softmax_scale is not None else q.shape[-1] ** -0.5
    return torch.stack([_sdpa(q[b], k[b], v[b], scale) for b in range(q.shape[0])])


def flash_attn_varlen_func(q, k, v, cu_seqlens_q, cu_seqlens_k, max_seqlen_q, max_seqlen_k, dropout_p=0.0, softmax_scale=None,
                           causal=False, **kw):
    """Packed sequences: q [total_q, H, D], k / v [total_k, H, D]; sample b owns rows cu_seqlens[b]:cu_seqlens[b+1]."""
    assert dropout_p == 0.0 and not causal
    scale = softmax_scale if softmax_scale is not None else q.shape[-1] ** -0.5
    out = torch.empty_like(q)
    cq, ck = cu_seqlens_q.tolist(), cu_seqlens_k.tolist()
    for b in range(len(cq) - 1):
        out[cq[b]:cq[b + 1]] = _sdpa(q[cq[b]:cq[b + 1]], k[ck[b]:ck[b + 1]], v[ck[b]:ck[b + 1]], scale)
    return out
