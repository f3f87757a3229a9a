"""GPU: LongCat-Video DiT kernels (csrc/longcat_ops.hip) and the whole HIP forward (worldforge_amd/longcat_dit.py) against the CPU oracle
(oracle/longcat_dit.py, pinned to the imported reference) and against the reference's own golden outputs (tests/golden/g11).

Tolerances (stated): the element-wise kernels reproduce the reference's bf16 rounding points, so they must match a torch
restatement with the same casts to <= 1 bf16 ulp (2^-7 relative); whole model: relative L2 error <= 2e-2 against the fp32 oracle on
identical bf16-valued weights and inputs (bf16 activations between layers, as the reference's bf16 model)."""
import os

import numpy as np
import pytest
import torch

from oracle import longcat_dit as olc

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF, F32 = torch.bfloat16, torch.float32
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "g11_longcat_dit.npz"))


def _rand(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


def _rel_l2(got, want):
    got, want = got.float().cpu(), want.float().cpu()
    return ((got - want).norm() / (want.norm() + 1e-12)).item()


def _ulp_close(got, want, ulps=1.0):
    got, want = got.float().cpu(), want.float().cpu()
    tol = ulps * 2.0 ** -7 * want.abs().clamp_min(1e-3)
    bad = (got - want).abs() > tol
    assert not bad.any(), ((got - want).abs().max().item(), int(bad.sum()))


@pytest.mark.parametrize("L,C,tpf", [(24, 256, 8), (30, 4096, 10), (7, 8192, 7), (12, 2048, 4)])
@pytest.mark.parametrize("affine", [False, True])
def test_ln_modulate(L, C, tpf, affine):
    from worldforge_amd._ffi import call
    from worldforge_amd import ops
    T = L // tpf
    x = _rand((L, C), 1, 2.0).to(BF)
    out = torch.full((L, C), float("nan"), dtype=BF, device=DEV)
    xd = x.to(DEV)
    if affine:
        w, b = (_rand((C,), 2, 0.1) + 1).to(DEV), _rand((C,), 3, 0.1).to(DEV)
        call("wf_lc_ln_modulate", xd.data_ptr(), w.data_ptr(), b.data_ptr(), 0, 0, 0, None, 0, out.data_ptr(), L, C, 1e-6, ops.stream())
        want = olc.layer_norm(x, w.cpu(), b.cpu())
    else:
        mod = _rand((T, 3 * C), 4, 0.3).to(DEV)  # [shift | scale | unused] rows 3C apart
        shift, scale = mod[:, :C], mod[:, C:2 * C]
        call("wf_lc_ln_modulate", xd.data_ptr(), scale.data_ptr(), shift.data_ptr(), mod.stride(0), tpf, 0, None, 1, out.data_ptr(), L, C, 1e-6,
             ops.stream())
        want = olc.modulate(x, shift.cpu(), scale.cpu(), tpf)
    assert torch.isfinite(out.float()).all()
    _ulp_close(out, want)


@pytest.mark.parametrize("L,C,tpf,gated", [(24, 256, 8, True), (30, 4096, 10, True), (9, 384, 3, False)])
def test_gate_residual(L, C, tpf, gated):
    from worldforge_amd._ffi import call
    from worldforge_amd import ops
    x, y = _rand((L, C), 1).to(BF), _rand((L, 2 * C), 2).to(BF)
    gate = _rand((L // tpf, 2 * C), 3).to(DEV)
    xd, yd = x.to(DEV), y.to(DEV)
    yv = yd[:, C:]
    call("wf_lc_gate_residual", xd.data_ptr(), yv.data_ptr(), yd.stride(0), gate[:, C:].data_ptr() if gated else None, gate.stride(0),
         tpf if gated else 0, 0, None, L, C, ops.stream())
    g = gate[:, C:].cpu().repeat_interleave(tpf, dim=0) if gated else 1.0
    want = (x.float() + g * y[:, C:].float()).to(BF)
    assert torch.equal(xd.cpu(), want)


@pytest.mark.parametrize("L,H,rope", [(50, 2, True), (33, 32, True), (20, 3, False)])
def test_norm_heads(L, H, rope):
    from worldforge_amd._ffi import call
    from worldforge_amd import ops
    from worldforge_amd.longcat_dit import rope_tables
    C = H * 128
    src = _rand((L, 3 * C), 1, 1.5).to(BF)
    w = (_rand((128,), 2, 0.05) + 1).to(BF).float()
    f, h, wd = (L // 10, 2, 5) if rope else (1, 1, L)
    Lr = f * h * wd
    cos, sin = rope_tables(128, f, h, wd)
    out = torch.zeros((H, Lr + 3, 128), dtype=BF, device=DEV)
    sd = src.to(DEV)
    view = sd[:Lr, C:2 * C]
    wd_, cd, sn = w.to(DEV), cos.to(DEV), sin.to(DEV)  # held: a temporary would be freed before the kernel reads it
    call("wf_lc_norm_heads", view.data_ptr(), sd.stride(0), wd_.data_ptr(), cd.data_ptr() if rope else None,
         sn.data_ptr() if rope else None, out.data_ptr(), Lr, Lr + 3, H, 1e-6, 1.0, ops.stream())
    # out_scale folds the attention's softmax_scale * log2(e) in front of the one rounding: == bf16(scale * unrounded result) to 1 ulp
    sc = 1.4426950408889634 / 128 ** 0.5
    out_s = torch.zeros((H, Lr + 3, 128), dtype=BF, device=DEV)
    call("wf_lc_norm_heads", view.data_ptr(), sd.stride(0), wd_.data_ptr(), cd.data_ptr() if rope else None,
         sn.data_ptr() if rope else None, out_s.data_ptr(), Lr, Lr + 3, H, 1e-6, sc, ops.stream())
    ref_s = out[:, :Lr].float().cpu() * sc
    assert ((out_s[:, :Lr].float().cpu() - ref_s).abs() <= 2 ** -7 * ref_s.abs() + 1e-30).all()   # two bf16 roundings apart at most
    q = src[:Lr, C:2 * C].view(Lr, H, 128).permute(1, 0, 2)  # [H, L, D] bf16
    want = olc.rms_norm_head(q, w.to(BF))
    if rope:
        want = olc.rope_apply(want, olc.rope_angles(128, f, h, wd))
        # the pair table of the product equals the reference's repeated table
        assert torch.equal(olc.rope_angles(128, f, h, wd)[:, 0::2].cos(), cos)
    _ulp_close(out[:, :Lr], want)
    assert out[:, Lr:].abs().max().item() == 0


@pytest.mark.parametrize("L,Hd", [(17, 768), (40, 11008)])
def test_swiglu(L, Hd):
    from worldforge_amd._ffi import call
    from worldforge_amd import ops
    a = _rand((L, 2 * Hd), 1, 2.0).to(BF)
    out = torch.empty((L, Hd), dtype=BF, device=DEV)
    ad = a.to(DEV)
    call("wf_lc_swiglu", ad.data_ptr(), ad.stride(0), out.data_ptr(), L, Hd, ops.stream())
    want = torch.nn.functional.silu(a[:, :Hd]) * a[:, Hd:]
    _ulp_close(out, want)


def _cfg(C, heads, depth, cap, ct, zpad=False):
    from worldforge_amd.longcat_dit import LongCatConfig
    return (LongCatConfig(hidden_size=C, depth=depth, num_heads=heads, caption_channels=cap, adaln_tembed_dim=ct, text_tokens_zero_pad=zpad),
            olc.LongCatConfig(hidden_size=C, depth=depth, num_heads=heads, caption_channels=cap, adaln_tembed_dim=ct, text_tokens_zero_pad=zpad))


@pytest.mark.parametrize("name", ["tiny", "odd", "zpad"])
def test_forward_matches_reference_goldens(name):
    """The reference's own outputs (fp32 model, CFG batch of two with caption masks): each sample through the HIP forward."""
    from worldforge_amd.longcat_dit import LongCatVideoTransformer3DModel
    C, heads, depth, cap, ct, ncond, zpad = (int(v) for v in G[f"{name}_cfg"])
    if C // heads != 128:
        pytest.skip("head_dim 128 only")
    cfg, ocfg = _cfg(C, heads, depth, cap, ct, bool(zpad))
    m = LongCatVideoTransformer3DModel(cfg, DEV).load_state_dict(olc.random_weights(ocfg, seed=21))
    x = torch.from_numpy(G[f"{name}_x"]).to(BF).to(DEV)
    for b in range(2):
        got = m.forward_tokens(x, G[f"{name}_ts"][b].tolist(), torch.from_numpy(G[f"{name}_cap"][b]).to(BF).to(DEV),
                               torch.from_numpy(G[f"{name}_mask"][b]), ncond)
        assert torch.isfinite(got).all()
        assert _rel_l2(got, torch.from_numpy(G[f"{name}_out"][b])) <= 2e-2, (name, b)


@pytest.mark.parametrize("C,heads,depth,T,h,w,ncond", [(256, 2, 3, 4, 8, 12, 1), (512, 4, 2, 3, 10, 6, 0), (384, 3, 2, 5, 6, 22, 2)])
def test_forward_matches_oracle(C, heads, depth, T, h, w, ncond):
    from worldforge_amd.longcat_dit import LongCatVideoTransformer3DModel
    cfg, ocfg = _cfg(C, heads, depth, 96, 64)
    W = olc.random_weights(ocfg, seed=4)
    m = LongCatVideoTransformer3DModel(cfg, DEV).load_state_dict(W)
    x = _rand((16, T, h, w), 11).to(BF)
    cap = _rand((40, 96), 12).to(BF)
    mask = torch.zeros(40, dtype=torch.int64)
    mask[:29] = 1
    ts = [0.0] * ncond + [812.0] * (T - ncond)
    want = olc.forward(W, ocfg, x.float(), torch.tensor(ts), cap.float(), mask, num_cond_latents=ncond)
    got = m.forward_tokens(x.to(DEV), ts, cap.to(DEV), mask, ncond)
    assert _rel_l2(got, want) <= 2e-2, _rel_l2(got, want)
    # the diffusers-style call: batch of two (negative, positive) with per-frame timesteps, as pipeline_longcat_video.py:857-873
    out = m(torch.stack([x, x]).to(DEV), torch.tensor([ts, ts]), torch.stack([cap, cap])[:, None].to(DEV), torch.stack([mask, mask]),
            num_cond_latents=ncond)
    assert out.shape == (2, 16, T, h, w) and out.dtype == F32
    assert torch.equal(out[0], out[1]) and torch.equal(out[0], got)


def test_call_rounds_timesteps_to_model_dtype_and_rejects_kv_cache():
    from worldforge_amd.longcat_dit import LongCatVideoTransformer3DModel
    cfg, ocfg = _cfg(256, 2, 1, 64, 64)
    m = LongCatVideoTransformer3DModel(cfg, DEV).load_state_dict(olc.random_weights(ocfg, seed=1))
    x = _rand((1, 16, 2, 4, 4), 1).to(BF).to(DEV)
    cap = _rand((1, 1, 8, 64), 2).to(BF).to(DEV)
    a = m(x, torch.tensor([637.0]), cap)                     # bf16(637) = 636  (longcat_video_dit.py:304-306)
    b = m(x, torch.tensor([[636.0, 636.0]]), cap)
    assert torch.equal(a, b)
    with pytest.raises(NotImplementedError):
        m(x, torch.tensor([1.0]), cap, return_kv=True)


def test_full_width_block_matches_oracle():
    """One block at the released width (hidden 4096, 32 heads, SwiGLU 11008, caption 4096, AdaLN 512) on a short clip."""
    from worldforge_amd.longcat_dit import LongCatVideoTransformer3DModel
    cfg, ocfg = _cfg(4096, 32, 1, 4096, 512)
    assert cfg.ffn_hidden == 11008
    W = olc.random_weights(ocfg, seed=8)
    m = LongCatVideoTransformer3DModel(cfg, DEV).load_state_dict(W)
    T, h, w = 3, 16, 20
    x = _rand((16, T, h, w), 31).to(BF)
    cap = _rand((64, 4096), 32).to(BF)
    mask = torch.zeros(64, dtype=torch.int64)
    mask[:50] = 1
    ts = [0.0, 500.0, 500.0]
    want = olc.forward(W, ocfg, x.float(), torch.tensor(ts), cap.float(), mask, num_cond_latents=1)
    got = m.forward_tokens(x.to(DEV), ts, cap.to(DEV), mask, 1)
    assert _rel_l2(got, want) <= 2e-2, _rel_l2(got, want)


def test_lora_fold_matches_reference_runtime_lora():
    """fold_lora at load vs the reference's run-time LoRA (golden g13), through the HIP forward."""
    from tests.fakes import lora_state
    from worldforge_amd.longcat_dit import LongCatVideoTransformer3DModel, fold_lora
    L = np.load(os.path.join(os.path.dirname(__file__), "golden", "g13_longcat_lora.npz"))
    cfg, ocfg = _cfg(256, 2, 2, 64, 64)
    W = olc.random_weights(ocfg, seed=21)
    m = LongCatVideoTransformer3DModel(cfg, DEV).load_state_dict(fold_lora(W, lora_state(ocfg), multiplier=0.8, network_dim=8,
                                                                           network_alpha=4))
    got = m.forward_tokens(torch.from_numpy(L["x"]).to(BF).to(DEV), L["ts"].tolist(), torch.from_numpy(L["cap"]).to(BF).to(DEV),
                           torch.from_numpy(L["mask"]), 1)
    assert _rel_l2(got, torch.from_numpy(L["out"])) <= 2e-2
    assert _rel_l2(got, torch.from_numpy(L["out_base"])) > 5e-2   # and not the base model
    with pytest.raises(KeyError):
        fold_lora(W, {"lora___lorahyphen___blocks___lorahyphen___9___lorahyphen___attn___lorahyphen___proj.lora_down.weight": torch.zeros(8, 256)})


def test_row_offset_of_a_token_shard():
    """row0: the per-frame parameters are selected by the GLOBAL token index when the rows are one rank's shard."""
    from worldforge_amd._ffi import call
    from worldforge_amd import ops
    L, C, tpf, lo = 40, 256, 8, 13
    x, y = _rand((L, C), 1, 2.0).to(BF), _rand((L, C), 2).to(BF)
    mod = _rand((L // tpf, 3 * C), 4, 0.3).to(DEV)
    shift, scale, gate = mod[:, :C], mod[:, C:2 * C], mod[:, 2 * C:]
    full = torch.empty((L, C), dtype=BF, device=DEV)
    part = torch.empty((L - lo, C), dtype=BF, device=DEV)
    xd = x.to(DEV)
    call("wf_lc_ln_modulate", xd.data_ptr(), scale.data_ptr(), shift.data_ptr(), mod.stride(0), tpf, 0, None, 1, full.data_ptr(), L, C, 1e-6, ops.stream())
    call("wf_lc_ln_modulate", xd[lo:].data_ptr(), scale.data_ptr(), shift.data_ptr(), mod.stride(0), tpf, lo, None, 1, part.data_ptr(), L - lo, C, 1e-6,
         ops.stream())
    assert torch.equal(full[lo:], part)
    xa, xb, yd = x.to(DEV).clone(), x.to(DEV).clone(), y.to(DEV)
    call("wf_lc_gate_residual", xa.data_ptr(), yd.data_ptr(), yd.stride(0), gate.data_ptr(), mod.stride(0), tpf, 0, None, L, C, ops.stream())
    call("wf_lc_gate_residual", xb[lo:].data_ptr(), yd[lo:].data_ptr(), yd.stride(0), gate.data_ptr(), mod.stride(0), tpf, lo, None, L - lo, C, ops.stream())
    assert torch.equal(xa[lo:], xb[lo:])


def test_group_index_replaces_the_frame_division():
    """group_index: rows in an arbitrary (here shuffled) order pick their frame's parameters through a per-row index."""
    from worldforge_amd._ffi import call
    from worldforge_amd import ops
    L, C, tpf = 48, 256, 8
    g = torch.Generator().manual_seed(5)
    perm = torch.randperm(L, generator=g)
    x, y = _rand((L, C), 1, 2.0).to(BF), _rand((L, C), 2).to(BF)
    mod = _rand((L // tpf, 3 * C), 4, 0.3).to(DEV)
    shift, scale, gate = mod[:, :C], mod[:, C:2 * C], mod[:, 2 * C:]
    gi = (perm // tpf).to(torch.int32).to(DEV)
    a = torch.empty((L, C), dtype=BF, device=DEV)
    b = torch.empty((L, C), dtype=BF, device=DEV)
    xd, xp = x.to(DEV), x[perm].to(DEV).contiguous()
    call("wf_lc_ln_modulate", xd.data_ptr(), scale.data_ptr(), shift.data_ptr(), mod.stride(0), tpf, 0, None, 1, a.data_ptr(), L, C, 1e-6, ops.stream())
    call("wf_lc_ln_modulate", xp.data_ptr(), scale.data_ptr(), shift.data_ptr(), mod.stride(0), 0, 0, gi.data_ptr(), 1, b.data_ptr(), L, C, 1e-6, ops.stream())
    assert torch.equal(a[perm.to(DEV)], b)
    xa, xb, yd, yp = xd.clone(), xp.clone(), y.to(DEV), y[perm].to(DEV).contiguous()
    call("wf_lc_gate_residual", xa.data_ptr(), yd.data_ptr(), yd.stride(0), gate.data_ptr(), mod.stride(0), tpf, 0, None, L, C, ops.stream())
    call("wf_lc_gate_residual", xb.data_ptr(), yp.data_ptr(), yp.stride(0), gate.data_ptr(), mod.stride(0), 0, 0, gi.data_ptr(), L, C, ops.stream())
    assert torch.equal(xa[perm.to(DEV)], xb)
