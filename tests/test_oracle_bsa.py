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


# ---- the reference's Triton kernel itself (g18: executed by Triton's interpreter on CPU tensors, tools/make_goldens.py bsa_triton) -------------
G18 = np.load(os.path.join(os.path.dirname(__file__), "golden", "g18_bsa_triton.npz"))


def _g18_inputs(name):
    from tests.fakes import bsa_triton_inputs
    q, k, v = bsa_triton_inputs(name)
    sums = np.array([q.double().sum().item(), k.double().sum().item(), v.double().sum().item()])
    assert np.allclose(sums, G18[f"{name}_insum"], rtol=0, atol=1e-6), "seeded inputs drifted from the ones the fixture was recorded on"
    return q, k, v


def test_sparse_attention_equals_the_reference_triton_kernel():
    """fp32 tensors: the restatement (masked dense softmax) == the kernel's online softmax over the selected blocks, to fp32 rounding."""
    from tests.fakes import BSA_TRITON_CASES
    for name in ("k128", "k64", "varlen"):
        c = BSA_TRITON_CASES[name]
        q, k, v = _g18_inputs(name)
        idx = torch.from_numpy(G18[f"{name}_idx"])
        lens = torch.from_numpy(G18[f"{name}_lens"]) if name == "varlen" else None
        got = obsa.sparse_attention(q, k, v, idx, c["block"], c["block"], 128 ** -0.5, lens)[:, ::4]
        assert np.abs(got.numpy() - G18[f"{name}_out"]).max() <= 2e-6, name
        if name != "varlen":  # the selection the reference made == the oracle's own (pooling + scores + top-k)
            own = obsa.select_topk(obsa.mean_pool(q, c["block"]), obsa.mean_pool(k, c["block"]), c["sparsity"])
            assert np.array_equal(own.numpy(), G18[f"{name}_idx"]) and (G18[f"{name}_lens"] == own.shape[-1]).all()
    # empty selections give exact zeros (acc = 0, l = 1 in the kernel), full-length rows the dense softmax
    lens = G18["varlen_lens"]
    out = G18["varlen_out"]
    blk_rows = 128 // 4
    for h in range(lens.shape[0]):
        for b in range(lens.shape[1]):
            if lens[h, b] == 0:
                assert not out[h, b * blk_rows:(b + 1) * blk_rows].any()


def test_half_tensors_round_p_before_the_value_product():
    """fp16 tensors (flash_attn_bsa_varlen_mask.py:259 `p.to(v.dtype)`): with p rounded to fp16 before P V and the row sums taken from the
    fp32 p the restatement lands within fp16 output rounding of the kernel; without the rounding it is measurably further away."""
    from tests.fakes import BSA_TRITON_CASES
    c = BSA_TRITON_CASES["half"]
    q, k, v = _g18_inputs("half")
    idx = torch.from_numpy(G18["half_idx"])
    want = G18["half_out"]
    got = obsa.sparse_attention(q, k, v, idx, c["block"], c["block"], 128 ** -0.5, p_dtype=torch.float16)[:, ::4].half().float().numpy()
    plain = obsa.sparse_attention(q, k, v, idx, c["block"], c["block"], 128 ** -0.5)[:, ::4].numpy()
    assert np.abs(got - want).max() <= 1.5e-3          # fp16 output spacing at |o| < 2 is 9.8e-4
    assert np.abs(got - want).mean() <= np.abs(plain - want).mean() * 1.05


def test_3d_block_interface_equals_reference():
    """flash_attn_bsa_3d (bsa_interface.py:612-659): 3D-block permute -> gating -> top-k -> sparse kernel -> inverse permute."""
    from tests.fakes import BSA_TRITON_CASES
    c = BSA_TRITON_CASES["thw"]
    q, k, v = _g18_inputs("thw")
    got = obsa.flash_attn_bsa_3d(q, k, v, c["grid"], c["grid"], sparsity=c["sparsity"], chunk_q=c["chunk"], chunk_k=c["chunk"])[:, ::4]
    assert np.abs(got.numpy() - G18["thw_out"]).max() <= 2e-6


def test_cdf_counts_on_bf16_scores_equal_reference():
    """g14c: the reference's get_select_indices_cdf_from_score / _cdf_topk_from_score on BF16 scores (what its bf16 model hands them).
    oracle.bsa.cdf_counts_bf16 spells out every bf16 rounding point of that eager chain and reproduces all counts exactly; the fp32 rule
    (select_cdf on float scores) does NOT -- it is off by up to several blocks on most rows near 0.9."""
    C = np.load(os.path.join(os.path.dirname(__file__), "golden", "g14c_bsa_cdf_bf16.npz"))
    n_fp32_differs = 0
    for name in ("flat770", "mid770", "peaked770", "small96"):
        score = torch.from_numpy(C[f"{name}_score_bits"].view(np.int16).copy()).view(torch.bfloat16)
        for thr, sp in ((0.3, None), (0.5, None), (0.9, None), (0.95, None), (0.3, 0.75), (0.9, 0.875)):
            want = torch.from_numpy(C[f"{name}_lens_thr{thr}_sp{sp}"]).long()
            got = obsa.cdf_counts_bf16(score, thr, sp)
            assert torch.equal(got, want), (name, thr, sp, (got - want).abs().max())
            w = torch.softmax(score.float() / 128 ** 0.5, -1)
            n32 = (torch.cumsum(torch.sort(w, -1, descending=True).values, -1) <= thr).sum(-1)
            if sp is not None:
                n32 = n32.clamp_min(int((1 - sp) * score.shape[-1]))
            n_fp32_differs += int((n32 != want).sum())
    assert n_fp32_differs > 100          # the fp32 rule is a different selection: this golden is what tells them apart
