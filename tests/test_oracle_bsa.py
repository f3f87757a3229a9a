"""CPU: oracle/bsa.py against the reference's own BSA helper functions (tests/golden/g14_bsa.npz) and its internal consistency."""
import os

import numpy as np
import torch

from oracle import bsa as obsa

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "g14_bsa.npz"))


def test_block_permutation_equals_reference_rearrange():
    for name in ("a", "b"):
        T, H, W, t, h, w = (int(v) for v in G[f"{name}_shape"])
        perm = obsa.block_permutation(T, H, W, t, h, w)
        assert np.array_equal(perm.numpy(), G[f"{name}_perm"])
        assert sorted(perm.tolist()) == list(range(T * H * W))


def test_pooling_and_topk_selection_equal_reference():
    for name, dt in (("f32", torch.float32), ("bf16", torch.bfloat16)):
        Hh, Sq, Sk, blk, sp = (int(v) for v in G[f"{name}_cfg"])
        q, k = torch.from_numpy(G[f"{name}_q"]).to(dt), torch.from_numpy(G[f"{name}_k"]).to(dt)
        qc, kc = obsa.mean_pool(q, blk), obsa.mean_pool(k, blk)
        assert torch.equal(qc.float(), torch.from_numpy(G[f"{name}_qc"])) and torch.equal(kc.float(), torch.from_numpy(G[f"{name}_kc"]))
        idx = obsa.select_topk(qc, kc, sp / 1000.0)
        assert np.array_equal(idx.numpy(), G[f"{name}_idx"])
        assert (G[f"{name}_lens"] == idx.shape[-1]).all()


def test_sparse_attention_with_all_blocks_is_dense_and_3d_wrapper_round_trips():
    g = torch.Generator().manual_seed(1)
    Hh, T, H, W = 2, 4, 4, 8
    S = T * H * W
    q, k, v = (torch.randn(Hh, S, 128, generator=g) for _ in range(3))
    dense = torch.softmax(torch.einsum("hqd,hkd->hqk", q, k) * 128 ** -0.5, -1) @ v
    out = obsa.flash_attn_bsa_3d(q, k, v, (T, H, W), (T, H, W), sparsity=0.0, chunk_q=(2, 2, 4), chunk_k=(2, 2, 4))
    assert (out - dense).abs().max() < 1e-5
    sp, idx = obsa.flash_attn_bsa_3d(q, k, v, (T, H, W), (T, H, W), sparsity=0.75, chunk_q=(2, 2, 4), chunk_k=(2, 2, 4), return_indices=True)
    assert idx.shape == (Hh, 8, 2) and (sp - dense).abs().max() > 1e-3


def test_cdf_selection_equals_reference():
    C = np.load(os.path.join(os.path.dirname(__file__), "golden", "g14b_bsa_cdf.npz"))
    for name in ("cdf", "cdf_topk"):
        thr, sp = (float(v) for v in C[f"{name}_cfg"])
        idx, lens = obsa.select_cdf(torch.from_numpy(C[f"{name}_qc"]), torch.from_numpy(C[f"{name}_kc"]), thr, None if sp < 0 else sp)
        assert np.array_equal(lens.numpy(), C[f"{name}_lens"])
        assert np.array_equal(idx.numpy(), C[f"{name}_idx"])
    # variable-length lists in the masked attention: only the first `lens` entries of a row count
    g = torch.Generator().manual_seed(2)
    q, k, v = (torch.randn(1, 256, 128, generator=g) for _ in range(3))
    idx = torch.tensor([[[1, 0], [0, 1]]])
    a = obsa.sparse_attention(q, k, v, idx, 128, 128, 0.1, block_lens=torch.tensor([[1, 2]]))
    b = obsa.sparse_attention(q, k, v, torch.tensor([[[1], [0]]]), 128, 128, 0.1)
    assert torch.equal(a[:, :128], b[:, :128]) and not torch.equal(a[:, 128:], b[:, 128:])
