"""GPU: BASELINE config 3 (Wan2.1-I2V-14B-720P, 81 frames of 720 x 1280 -> latents [1,16,21,90,160], L = 75 600 tokens) -- the sizes
at which "attention tile sizing at 720p" matters: 296 query blocks of 256 rows per head (the last one ragged: 75 600 = 295 * 256 + 80),
1182 key tiles of 64 (the last one ragged: 75 600 = 1181 * 64 + 16), 11 840 workgroups per launch.  Same method as
test_gpu_fullsize.py: sampled rows against the oracle, whole-tensor properties, shard / lock-step invariance, sharded == unsharded VAE."""
import math
import threading

import pytest
import torch

from oracle import dit as odit

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
BF, F32 = torch.bfloat16, torch.float32
L_C3, H_C3 = 75600, 40
T3, h3, w3 = 21, 90, 160


def _dev_randn(shape, seed, scale=1.0, dtype=BF):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return (torch.randn(shape, generator=g, device=DEV, dtype=F32) * scale).to(dtype)


def test_self_attention_c3_sampled_rows_vs_oracle_normalisation_and_key_padding():
    from tests.test_gpu_fullsize import _attn_inputs
    from worldforge_amd import dit
    L, H = L_C3, H_C3
    q, k, v, vt = _attn_inputs(L, H, 700)
    out = torch.empty((L, H * 128), dtype=BF, device=DEV)
    dit.attention(q, k, vt, out, L, 1.0 / math.sqrt(128.0))
    torch.cuda.synchronize()
    rows = [0, 1, 255, 256, 4096, 32759, 32760, 65535, 65536, 75519, 75520, 75521, L - 81, L - 80, L - 17, L - 16, L - 2, L - 1]
    for h in (0, 15, 39):
        qs = q[h, rows].float().cpu().unsqueeze(1)
        want = odit.attention(qs, k[h, :L].float().cpu().unsqueeze(1), v[h].float().cpu().unsqueeze(1))[:, 0]
        got = out[rows, h * 128:(h + 1) * 128].float().cpu()
        err = (got - want).abs().max().item()
        assert err <= 2e-3, (h, err)
    vt.fill_(1.0)
    dit.attention(q, k, vt, out, L, 1.0 / math.sqrt(128.0))
    assert (out.float() - 1.0).abs().max().item() <= 2.0 ** -7
    k[:, L:] = 1e4
    out2 = torch.empty_like(out)
    dit.attention(q, k, vt, out2, L, 1.0 / math.sqrt(128.0))
    assert torch.equal(out, out2)


def test_dit_c3_tokens_two_layers_sharded_and_lockstep_match_single_rank():
    """Two real-width layers on the 75 600-token grid: lock-step CFG pair == two sequential forwards; 4 simulated ranks (shards of
    18 944 / 18 768 tokens, K / V^T all-gathered and consumed in place) == one rank, bit for bit."""
    from tests.fakes import SimComm
    from worldforge_amd import dit
    cfg = dit.DiTConfig.wan_i2v_14b()
    cfg.num_layers = 2
    x = _dev_randn((36, T3, h3, w3), 710)
    ca, cb = _dev_randn((200, 4096), 711, 0.1), _dev_randn((60, 4096), 712, 0.1)
    clip = _dev_randn((257, 1280), 713)
    m0 = dit.WanTransformer3DModel(cfg, DEV).init_random(5)
    ref_a = m0.forward_tokens(x, 777.0, ca, clip).clone()
    ref_b = m0.forward_tokens(x, 777.0, cb, clip).clone()
    assert torch.isfinite(ref_a).all() and ref_a.abs().max().item() > 0
    a, b = m0.forward_tokens_pair(x, 777.0, ca, cb, clip, interleave=True)
    assert torch.equal(a, ref_a) and torch.equal(b, ref_b)
    m0._ws.clear()
    P = 4
    assert dit.kv_splits(cfg.num_heads, (-(-L_C3 // P) + 63) // 64 * 64, L_C3) == 1
    shared = {"slots": [None] * P, "bar": threading.Barrier(P)}
    res, errs = [None] * P, []

    def worker(r):
        try:
            m = dit.WanTransformer3DModel(cfg, DEV, comm=SimComm(P, r, shared))
            m.w = m0.w
            res[r] = tuple(t.clone() for t in m.forward_tokens_pair(x, 777.0, ca, cb, clip))
        except Exception as e:  # pragma: no cover
            errs.append(e)
            shared["bar"].abort()

    th = [threading.Thread(target=worker, args=(r,)) for r in range(P)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    for r in range(P):
        assert torch.equal(res[r][0], ref_a), (r, (res[r][0] - ref_a).abs().max())
        assert torch.equal(res[r][1], ref_b), r


def test_vae_c3_row_sharded_equals_unsharded():
    """81 x 720 x 1280: decode and encode on 4 simulated ranks (row slabs + halo all-gather) == unsharded, bit for bit."""
    from tests.fakes import SimComm
    from worldforge_amd.vae import AutoencoderKLWan
    v0 = AutoencoderKLWan(DEV, precision="bf16").init_random(seed=1)   # memory: the three-term operands of the fp32-class mode are 3x wider
    z = _dev_randn((1, 16, T3, h3, w3), 720, 1.0, F32)
    g = torch.Generator(device=DEV).manual_seed(721)
    video = torch.rand((1, 3, 81, 720, 1280), generator=g, device=DEV) * 2 - 1
    ref_dec = v0.decode(z, return_dict=False)[0]
    assert ref_dec.shape == (1, 3, 81, 720, 1280) and torch.isfinite(ref_dec).all()
    assert ref_dec.min().item() >= -1.0 and ref_dec.max().item() <= 1.0
    ref_mu = v0.encode(video).latent_dist.mode()
    assert ref_mu.shape == (1, 16, T3, h3, w3) and torch.isfinite(ref_mu).all()
    torch.cuda.empty_cache()   # the unsharded pass peaks at ~100 GB of activations; hand the cached blocks back before the rank threads
    P = 4
    shared = {"slots": [None] * P, "bar": threading.Barrier(P)}
    ok, errs = [False] * P, []

    def worker(r):
        try:
            m = AutoencoderKLWan(DEV, comm=SimComm(P, r, shared), precision="bf16")
            m.w = v0.w
            assert m.can_shard(h3)
            d = m.decode(z, return_dict=False)[0]
            mu = m.encode(video).latent_dist.mode()
            ok[r] = torch.equal(d, ref_dec) and torch.equal(mu, ref_mu)
        except Exception as e:  # pragma: no cover
            errs.append(e)
            shared["bar"].abort()

    th = [threading.Thread(target=worker, args=(r,)) for r in range(P)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    assert all(ok), ok
