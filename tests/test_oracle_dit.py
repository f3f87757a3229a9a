"""CPU: oracle DiT restatement against the golden recorded from the reference's in-tree WanModel (fp32, tiny configs)."""
import os

import numpy as np
import pytest
import torch

from oracle import dit as odit


@pytest.mark.parametrize("name", ["tiny", "odd"])
def test_dit_forward_matches_twin(name, golden_dir):
    g = np.load(os.path.join(golden_dir, "g7_dit.npz"))
    dim, heads, ffn, layers, T, h, w = g[f"{name}_cfg"].tolist()
    cfg = odit.DiTConfig(dim=dim, ffn_dim=ffn, num_heads=heads, num_layers=layers, text_dim=64)
    W = odit.random_weights(cfg, seed=11)
    out = odit.forward(W, cfg, torch.from_numpy(g[f"{name}_x"]), torch.tensor(749), torch.from_numpy(g[f"{name}_ctx"]),
                       torch.from_numpy(g[f"{name}_clip"]))
    want = g[f"{name}_out"]
    assert out.shape == want.shape
    err = np.abs(out.numpy() - want).max()
    print(name, "max abs err", err, "ref max", np.abs(want).max())
    # same fp32 arithmetic up to operation order inside SDPA / conv-vs-linear patch embedding
    assert err <= 2e-5 * max(1.0, np.abs(want).max())


def test_rope_table_layout():
    ang = odit.rope_tables(128, 3, 4, 5)
    assert ang.shape == (60, 64)
    # pair split 22 / 21 / 21 (model.py:478-485): first 22 pairs depend on the frame only
    assert torch.equal(ang[0, :22], ang[4 * 5 - 1, :22]) and not torch.equal(ang[0, :22], ang[4 * 5, :22])
    assert torch.equal(ang[0, 22:43], ang[4, 22:43]) and not torch.equal(ang[0, 22:43], ang[5, 22:43])
    assert not torch.equal(ang[0, 43:], ang[1, 43:])


def test_forward_rows_equals_forward_at_the_sampled_tokens():
    """oracle.dit.forward_rows (the row-sampled statement used for the full-token-count GPU check) against the pinned forward()."""
    cfg = odit.DiTConfig(dim=384, ffn_dim=640, num_heads=3, num_layers=1, text_dim=64)
    W = odit.random_weights(cfg, seed=13)
    g = torch.Generator().manual_seed(4)
    T, h, w = 3, 6, 10
    x = torch.randn(36, T, h, w, generator=g)
    ctx, clip = torch.randn(40, 64, generator=g), torch.randn(257, 1280, generator=g)
    full = odit.forward(W, cfg, x, torch.tensor(749), ctx, clip)
    rows = [0, 1, 14, 15, 29, 30, 44]
    got = odit.forward_rows(W, cfg, x, torch.tensor(749), ctx, clip, rows)
    want = odit.token_patches(full, cfg, rows)
    assert got.shape == want.shape == (len(rows), 64)
    assert (got - want).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item())
    with pytest.raises(AssertionError):
        odit.forward_rows(W, odit.DiTConfig(dim=384, ffn_dim=640, num_heads=3, num_layers=2, text_dim=64), x, torch.tensor(749), ctx, clip, rows)
