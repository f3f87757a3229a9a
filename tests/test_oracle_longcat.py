"""CPU: oracle/longcat_dit.py against outputs of the unmodified reference LongCatVideoTransformer3DModel (tests/golden/g11_longcat_dit.npz,
written by tools/make_goldens.py longcat).  The reference ran the CFG pair as a batch of two with caption masks; the oracle runs one
sample at a time, as the HIP path does."""
import os

import numpy as np
import pytest
import torch

from oracle import longcat_dit as olc

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "g11_longcat_dit.npz"))


def _case(name):
    C, heads, depth, cap, ct, ncond, zpad = (int(v) for v in G[f"{name}_cfg"])
    cfg = olc.LongCatConfig(hidden_size=C, depth=depth, num_heads=heads, caption_channels=cap, adaln_tembed_dim=ct,
                            text_tokens_zero_pad=bool(zpad))
    return cfg, ncond


@pytest.mark.parametrize("name", ["tiny", "odd", "zpad"])
def test_oracle_matches_reference_forward(name):
    cfg, ncond = _case(name)
    W = olc.random_weights(cfg, seed=21)
    x = torch.from_numpy(G[f"{name}_x"])
    for b in range(2):
        got = olc.forward(W, cfg, x, torch.from_numpy(G[f"{name}_ts"][b]), torch.from_numpy(G[f"{name}_cap"][b]),
                          torch.from_numpy(G[f"{name}_mask"][b]), num_cond_latents=ncond)
        want = torch.from_numpy(G[f"{name}_out"][b])
        err = (got - want).abs().max().item() / want.abs().max().item()
        assert err < 2e-5, (name, b, err)


def test_oracle_matches_reference_forward_with_the_references_own_triton_softmax():
    """g11b: the reference DiT with enable_bsa and sparsity 0 -- its self-attention is then a dense softmax computed
    by the reference's OWN Triton kernel (through Triton's interpreter), not by the flash-attn stand-in that served g11.  The oracle's
    dense forward (token order; the reference permutes to 3D-block order and back) must give the same velocities."""
    B = np.load(os.path.join(os.path.dirname(__file__), "golden", "g11b_longcat_dit_bsa0.npz"))
    C, heads, depth, cap, ct = (int(v) for v in B["cfg"])
    assert int(B["triton_calls"]) == depth
    cfg = olc.LongCatConfig(hidden_size=C, depth=depth, num_heads=heads, caption_channels=cap, adaln_tembed_dim=ct)
    W = olc.random_weights(cfg, seed=21)
    got = olc.forward(W, cfg, torch.from_numpy(B["x"]), torch.from_numpy(B["ts"][0]), torch.from_numpy(B["cap"][0]),
                      torch.from_numpy(B["mask"][0]), num_cond_latents=0)
    want = torch.from_numpy(B["out"][0])
    err = (got - want).abs().max().item() / want.abs().max().item()
    assert err < 2e-5, err


def test_ffn_width_and_rope_partition():
    cfg = olc.LongCatConfig()
    assert cfg.ffn_hidden == 11008
    ang = olc.rope_angles(128, 2, 3, 4)
    assert ang.shape == (24, 128)
    # 44 temporal, 42 row, 42 column entries; pairs share one frequency
    assert torch.equal(ang[:, 0::2], ang[:, 1::2])
    assert ang[0].abs().max() == 0 and ang[1, :44].abs().max() == 0 and ang[1, 86] == 1.0


def test_lora_fold_equals_reference_runtime_lora():
    """G13: the reference with a LoRA network enabled at run time (5 wrapped Linears per block, fused qkv / kv with separate up-blocks,
    multiplier 0.8, stored alpha_scale 0.5) == the oracle forward on the folded weights."""
    from tests.fakes import lora_state
    L = np.load(os.path.join(os.path.dirname(__file__), "golden", "g13_longcat_lora.npz"))
    cfg = olc.LongCatConfig(hidden_size=256, depth=2, num_heads=2, caption_channels=64, adaln_tembed_dim=64)
    W = olc.random_weights(cfg, seed=21)
    args = (torch.from_numpy(L["x"]), torch.from_numpy(L["ts"]), torch.from_numpy(L["cap"]), torch.from_numpy(L["mask"]))
    base = olc.forward(W, cfg, *args, num_cond_latents=1)
    assert (base - torch.from_numpy(L["out_base"])).abs().max() / np.abs(L["out_base"]).max() < 2e-5
    Wf = olc.fold_lora(W, lora_state(cfg), multiplier=0.8, network_dim=8, network_alpha=4)
    got = olc.forward(Wf, cfg, *args, num_cond_latents=1)
    want = torch.from_numpy(L["out"])
    assert (got - want).abs().max() / want.abs().max() < 2e-5
    assert (want - torch.from_numpy(L["out_base"])).abs().mean() > 0.05  # the LoRA really changes the output


@pytest.mark.parametrize("ncond,zpad", [(1, False), (0, False), (2, True)])
def test_forward_rows_equals_forward_at_the_sampled_tokens(ncond, zpad):
    """oracle.longcat_dit.forward_rows (the row-sampled statement used for the full-size GPU check) against the pinned forward()."""
    cfg = olc.LongCatConfig(hidden_size=256, depth=1, num_heads=2, caption_channels=64, adaln_tembed_dim=64, text_tokens_zero_pad=zpad)
    W = olc.random_weights(cfg, seed=5)
    g = torch.Generator().manual_seed(6)
    T, h, w = 4, 8, 12
    x = torch.randn(16, T, h, w, generator=g)
    cap = torch.randn(24, 64, generator=g)
    mask = torch.zeros(24, dtype=torch.int64)
    mask[:17] = 1
    ts = torch.full((T,), 637.0)
    ts[:ncond] = 0
    full = olc.forward(W, cfg, x, ts, cap, mask, num_cond_latents=ncond)
    tpf = (h // 2) * (w // 2)
    rows = [0, 1, tpf - 1, tpf, tpf + 5, 2 * tpf - 1, 2 * tpf, 4 * tpf - 1]
    got = olc.forward_rows(W, cfg, x, ts, cap, mask, ncond, rows)
    want = olc.token_patches(full, cfg, rows)
    assert got.shape == want.shape == (len(rows), 64)
    assert (got - want).abs().max().item() <= 2e-5 * max(1.0, want.abs().max().item())
