"""GPU: the hot-path kernels at the FULL sizes of BASELINE.json configs[1] (C2: 81 x 480 x 832 -> L = 32760 tokens, 40 heads,
d = 5120, FFN 13824), where a CPU oracle of the whole tensor would take hours.  Parity is established through
  * sampled rows / pixels recomputed on the CPU in fp32 from the same bf16 operands (oracle functions where they exist),
  * size-independent properties that hold for the whole tensor: exact power-of-two linearity, softmax normalisation,
    shard / lock-step invariance (bit-identical), full-tensor equality with the oracle for the HBM-bound element-wise ops."""
import math

import pytest
import torch

from oracle import dit as odit
from oracle import inject as oinject
from tests._tol import within

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
BF, F32 = torch.bfloat16, torch.float32
L_C2, H_C2, D_MODEL, D_FFN = 32760, 40, 5120, 13824


def _dev_randn(shape, seed, scale=1.0, dtype=BF):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return (torch.randn(shape, generator=g, device=DEV, dtype=F32) * scale).to(dtype)


# ---------------------------------------------------------------------------------------------------------------------
# self-attention core at C2 (a-17)
# ---------------------------------------------------------------------------------------------------------------------
def _attn_inputs(L, H, seed):
    Lp = (L + 63) // 64 * 64
    q = _dev_randn((H, L, 128), seed)
    k = torch.zeros((H, Lp, 128), dtype=BF, device=DEV)
    k[:, :L] = _dev_randn((H, L, 128), seed + 1)
    v = _dev_randn((H, L, 128), seed + 2)
    vt = torch.zeros((H, Lp // 64, 128, 64), dtype=BF, device=DEV)
    vpad = torch.zeros((H, Lp, 128), dtype=BF, device=DEV)
    vpad[:, :L] = v
    vt.copy_(vpad.view(H, Lp // 64, 64, 128).transpose(2, 3))
    return q, k, v, vt


def test_self_attention_c2_sampled_rows_vs_oracle_and_normalisation():
    from worldforge_amd import dit
    L, H = L_C2, H_C2
    q, k, v, vt = _attn_inputs(L, H, 100)
    out = torch.empty((L, H * 128), dtype=BF, device=DEV)
    dit.attention(q, k, vt, out, L, 1.0 / math.sqrt(128.0))
    torch.cuda.synchronize()
    # sampled (head, row) pairs incl. first / last rows, first / last heads, the ragged last 256-row block and last 64-key tile
    rows = [0, 1, 31, 32, 255, 256, 4095, 4096, 16383, 20000, L - 249, L - 65, L - 64, L - 2, L - 1]
    for h in (0, 7, 8, 23, 39):
        qs = q[h, rows].float().cpu().unsqueeze(1)                                  # [n, 1, 128]
        want = odit.attention(qs, k[h, :L].float().cpu().unsqueeze(1), v[h].float().cpu().unsqueeze(1))[:, 0]
        got = out[rows, h * 128:(h + 1) * 128].float().cpu()
        err = (got - want).abs().max().item()
        # outputs are O(1/sqrt(L)) averages of unit normals: |o| ~ 0.02; P is rounded to bf16 before P.V (as flash_attention)
        within("fullsize.attn_c2.max_abs", err, 2e-4)   # measured 1.0e-4
    # normalisation: with V = 1 every output is sum(p)/sum(p) = 1 up to the bf16 rounding of P (whole tensor)
    vt.fill_(1.0)
    dit.attention(q, k, vt, out, L, 1.0 / math.sqrt(128.0))
    dev = (out.float() - 1.0).abs().max().item()
    within("fullsize.attn_c2.ones", dev, 2.0 ** -9)   # measured 0 (exactly 1.0 everywhere); half a bf16 ulp allowed
    # scale invariance of the key padding: garbage in the padded key rows must not leak (mask by kv_len)
    k[:, L:] = 1e4
    out2 = torch.empty_like(out)
    dit.attention(q, k, vt, out2, L, 1.0 / math.sqrt(128.0))
    assert torch.equal(out, out2)


# ---------------------------------------------------------------------------------------------------------------------
# projections / FFN GEMMs at C2 (a-16)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,K,epi", [(3 * D_MODEL, D_MODEL, 0), (D_FFN, D_MODEL, 1), (D_MODEL, D_FFN, 2)])
def test_gemm_c2_sampled_rows_and_exact_linearity(N, K, epi):
    from worldforge_amd import dit
    M = L_C2
    x = _dev_randn((M, K), 200)
    w = _dev_randn((N, K), 201, 1.0 / math.sqrt(K))
    b = _dev_randn((N,), 202, 0.1, F32)
    odt = F32 if epi == 2 else BF
    out = torch.empty((M, N), dtype=odt, device=DEV)
    dit.gemm(x, w, b, out, epi)
    rows = [0, 1, 127, 128, 255, 256, 511, 512, 9999, M - 257, M - 256, M - 129, M - 2, M - 1]  # incl. the ragged last tile
    want = x[rows].float().cpu() @ w.float().cpu().T + b.cpu()
    if epi == 1:
        want = torch.nn.functional.gelu(want, approximate="tanh")
    got = out[rows].float().cpu()
    # bf16 outputs: one bf16 ulp of the largest output (measured 0.76 of it); fp32 outputs: measured 4.6e-6 at K = 13 824 (fp32 accumulation order)
    tol = 8e-8 * math.sqrt(K) if epi == 2 else 2.0 ** -8 * max(1.0, want.abs().max().item())
    within(f"fullsize.gemm.epi{epi}", (got - want).abs().max().item() / tol, 1.0)
    if epi != 1:
        # power-of-two linearity is exact in bf16 / fp32 (no bias): the whole [M, N] output, every tile and edge
        o1 = torch.empty((M, N), dtype=odt, device=DEV)
        o2 = torch.empty((M, N), dtype=odt, device=DEV)
        dit.gemm(x, w, None, o1, epi)
        dit.gemm(x * 4.0, w, None, o2, epi)
        assert torch.equal(o2, o1 * 4.0)
        assert torch.isfinite(o1.float()).all()


# ---------------------------------------------------------------------------------------------------------------------
# VAE convolution at the C2 pixel grid (a-21 / a-22): the 96 -> 96 3x3x3 layer over 81 x 480 x 832 pixels
# ---------------------------------------------------------------------------------------------------------------------
def test_conv3d_c2_sampled_pixels_borders_and_exact_linearity():
    from worldforge_amd import _ffi, ops
    T, H, W, C = 81, 480, 832, 96
    x = _dev_randn((T, H, W, C), 300)
    w = _dev_randn((C, C, 3, 3, 3), 301, 1.0 / math.sqrt(27 * C))
    b = _dev_randn((C,), 302, 0.1, F32)
    wk = w.permute(0, 2, 3, 4, 1).reshape(C, 27, C).contiguous()
    zp = torch.zeros(4096, dtype=BF, device=DEV)
    of = torch.empty((T, H, W, C), dtype=F32, device=DEV)

    wp = torch.empty((27, C // 16, C, 16), dtype=BF, device=DEV)
    _ffi.call("wf_conv3d_pack333", wk.data_ptr(), wp.data_ptr(), C, C, ops.stream())

    def run(inp, bias, dst):   # the LDS-resident patch kernel the VAE uses for this layer
        _ffi.call("wf_conv3d_333", inp.data_ptr(), wp.data_ptr(), bias.data_ptr() if bias is not None else None, None, dst.data_ptr(),
                  None, T, H, W, C, H, C, 1, zp.data_ptr(), zp.numel() * 2, 0, C, ops.stream())

    run(x, b, of)
    pts = [(0, 0, 0), (0, 0, W - 1), (0, H - 1, 0), (T - 1, H - 1, W - 1), (1, 1, 1), (2, 0, 5), (40, 239, 415), (80, 479, 0),
           (80, 0, 831), (17, 479, 831), (1, 0, 0), (63, 300, 511), (63, 300, 512)]
    wf = w.float().cpu()
    for (t, y, xx) in pts:
        acc = b.cpu().clone().double()
        for dt in range(3):
            tt = t + dt - 2                                   # causal: taps at t-2, t-1, t (vae.py:28-34)
            if tt < 0:
                continue
            for dy in range(3):
                yy = y + dy - 1
                if yy < 0 or yy >= H:
                    continue
                for dx in range(3):
                    xc = xx + dx - 1
                    if xc < 0 or xc >= W:
                        continue
                    acc += (wf[:, :, dt, dy, dx].double() @ x[tt, yy, xc].float().cpu().double())
        got = of[t, y, xx].cpu().double()
        within("fullsize.conv_c2.max_abs", (got - acc).abs().max().item(), 2.1e-6)   # measured 1.04e-6
    o1 = torch.empty_like(of)
    run(x, None, o1)
    x.mul_(2.0)
    run(x, None, of)
    o1.mul_(2.0)
    assert torch.equal(of, o1)


# ---------------------------------------------------------------------------------------------------------------------
# injection / scheduler element-wise set at the C2 pixel and latent sizes: the whole tensor against the oracle (a-4 .. a-12)
# ---------------------------------------------------------------------------------------------------------------------
def test_blend_and_latent_ops_c2_equal_oracle_on_the_whole_tensor():
    from worldforge_amd import ops
    g = torch.Generator().manual_seed(400)
    F_, H, W = 81, 480, 832
    ref = torch.rand((1, 3, F_, H, W), generator=g)
    mask = (torch.rand((1, 1, F_, H, W), generator=g) > 0.4).float() * torch.rand((1, 1, F_, H, W), generator=g)
    dec = torch.rand((1, 3, F_, H, W), generator=g) * 2 - 1
    got = ops.blend_pixels(ref.to(DEV), mask.to(DEV), dec.to(DEV)).cpu()
    assert torch.equal(got, oinject.blend_pixels(ref, mask, dec))
    lat = (1, 16, 21, 60, 104)
    a, b = torch.randn(lat, generator=g).to(BF), torch.randn(lat, generator=g).to(BF)
    assert torch.equal(ops.cfg_combine(a.to(DEV), b.to(DEV), 4.0).cpu(), oinject.cfg_combine(a, b, 4.0))
    want = oinject.dsg(a, b, 4.0)
    have = ops.dsg(a.to(DEV), b.to(DEV), 4.0).cpu()
    # the three global sums are reduced in a different (fixed) order than torch's: <= 1 bf16 ulp on values at a rounding boundary
    within("fullsize.dsg", (have.float() - want.float()).abs().max().item() / want.float().abs().max().item(), 1.1e-3)   # measured 5.6e-4
    assert (have != want).float().mean().item() < 5e-3


# ---------------------------------------------------------------------------------------------------------------------
# the DiT forward at the C2 token count (2 real-width layers): shard and lock-step invariance, bit for bit
# ---------------------------------------------------------------------------------------------------------------------
def test_dit_c2_tokens_sharded_and_lockstep_forwards_match_single_rank():
    import threading
    from tests.fakes import SimComm
    from worldforge_amd import dit
    cfg = dit.DiTConfig.wan_i2v_14b()
    cfg.num_layers = 2
    T, Hh, Ww = 21, 60, 104
    x = _dev_randn((36, T, Hh, Ww), 500)
    ca, cb = _dev_randn((200, 4096), 501, 0.1), _dev_randn((60, 4096), 502, 0.1)
    clip = _dev_randn((257, 1280), 503)
    m0 = dit.WanTransformer3DModel(cfg, DEV).init_random(5)
    ref_a = m0.forward_tokens(x, 777.0, ca, clip).clone()
    ref_b = m0.forward_tokens(x, 777.0, cb, clip).clone()
    assert torch.isfinite(ref_a).all() and ref_a.abs().max().item() > 0
    a, b = m0.forward_tokens_pair(x, 777.0, ca, cb, clip, interleave=True)
    assert torch.equal(a, ref_a) and torch.equal(b, ref_b)
    m0._ws.clear()
    # P = 4: shards of 8192 tokens (last one 8184), one KV sweep per launch -> bit-identical to the single-rank forward.
    # P = 8: shards of 4096 tokens (last one 4088): the attention launches split the KV sweep in two to fill the chip
    #        (dit.kv_splits), which re-associates the fp32 sums -> equal up to bf16-level rounding.
    for P in (4, 8):
        Lq = -(-(T * (Hh // 2) * (Ww // 2)) // P)
        Lq = (Lq + 63) // 64 * 64
        split = dit.kv_splits(cfg.num_heads, Lq, T * (Hh // 2) * (Ww // 2))
        assert split == (2 if P == 8 else 1)
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
            if split == 1:
                assert torch.equal(res[r][0], ref_a), (P, r, (res[r][0] - ref_a).abs().max())
                assert torch.equal(res[r][1], ref_b), (P, r)
            else:
                assert torch.equal(res[r][0], res[0][0]) and torch.equal(res[r][1], res[0][1])   # all ranks agree exactly
                within(f"fullsize.dit_split_kv.P{P}.a", (res[r][0] - ref_a).abs().max().item() / ref_a.abs().max().item(), 3.5e-3)   # measured 1.78e-3 / 1.75e-3
                within(f"fullsize.dit_split_kv.P{P}.b", (res[r][1] - ref_b).abs().max().item() / ref_a.abs().max().item(), 3.5e-3)


# ---------------------------------------------------------------------------------------------------------------------
# the VAE at the C2 video size: row-sharded (8 simulated ranks) == unsharded, bit for bit, for decode and encode (a-21 / a-22)
# ---------------------------------------------------------------------------------------------------------------------
def test_vae_c2_row_sharded_equals_unsharded():
    import threading
    from tests.fakes import SimComm
    from worldforge_amd.vae import AutoencoderKLWan
    v0 = AutoencoderKLWan(DEV, precision="bf16").init_random(seed=1)   # sharding invariance does not depend on the operand split; the fp32-class mode is covered in test_gpu_vae.py
    z = _dev_randn((1, 16, 21, 60, 104), 600, 1.0, F32)
    g = torch.Generator(device=DEV).manual_seed(601)
    video = torch.rand((1, 3, 81, 480, 832), generator=g, device=DEV) * 2 - 1
    ref_dec = v0.decode(z, return_dict=False)[0]
    assert ref_dec.shape == (1, 3, 81, 480, 832) and torch.isfinite(ref_dec).all()
    assert ref_dec.min().item() >= -1.0 and ref_dec.max().item() <= 1.0          # autoencoder_kl_wan.py:1222 clamp
    ref_mu = v0.encode(video).latent_dist.mode()
    assert ref_mu.shape == (1, 16, 21, 60, 104) and torch.isfinite(ref_mu).all()
    P = 8
    shared = {"slots": [None] * P, "bar": threading.Barrier(P)}
    ok, errs = [False] * P, []

    def worker(r):
        try:
            m = AutoencoderKLWan(DEV, comm=SimComm(P, r, shared), precision="bf16")
            m.w = v0.w
            assert m.can_shard(60)
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


# ---------------------------------------------------------------------------------------------------------------------
# the DiT at its real width (d = 5120, 40 heads, FFN 13824, text 4096) against the CPU oracle: two layers on the C1 token grid
# ---------------------------------------------------------------------------------------------------------------------
def test_dit_full_width_two_layers_vs_oracle():
    """Every projection at its production K (5120 / 13824) and all 40 heads, on the tokens of BASELINE configs[0] (9 frames of
    464 x 832 -> L = 4524): the HIP forward (bf16 MFMA operands, fp32 accumulation / residual stream) against the fp32 oracle run on
    the same bf16-rounded weights and inputs."""
    from worldforge_amd import dit
    ocfg = odit.DiTConfig(num_layers=2)
    assert (ocfg.dim, ocfg.num_heads, ocfg.ffn_dim, ocfg.text_dim) == (5120, 40, 13824, 4096)
    W = odit.random_weights(ocfg, seed=21)
    Wb = {k: (v.to(BF).float() if v.dim() >= 2 else v) for k, v in W.items()}
    cfg = dit.DiTConfig.wan_i2v_14b()
    cfg.num_layers = 2
    model = dit.WanTransformer3DModel(cfg, DEV).load_state_dict(W)
    g = torch.Generator().manual_seed(22)
    T, Hh, Ww = 3, 58, 104
    x = torch.randn(36, T, Hh, Ww, generator=g).to(BF)
    ctx = (torch.randn(200, 4096, generator=g) * 0.1).to(BF)
    clip = torch.randn(257, 1280, generator=g).to(BF)
    out = model.forward_tokens(x.to(DEV), 749.0, ctx.to(DEV), clip.to(DEV)).cpu()
    with torch.no_grad():
        orc = odit.forward(Wb, ocfg, x.float(), torch.tensor(749), ctx.float(), clip.float())
    rel = (out - orc).norm().item() / orc.norm().item()
    print(f"full-width DiT (2 layers, L = {T * (Hh // 2) * (Ww // 2)}): rel L2 vs oracle {rel:.3e}, max abs {(out - orc).abs().max().item():.3e}")
    assert torch.isfinite(out).all()
    within("fullsize.dit_2layers.rel_l2", rel, 4.7e-3)   # measured 2.33e-3
