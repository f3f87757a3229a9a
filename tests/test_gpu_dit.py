"""GPU: DiT kernels (MFMA GEMM, fused attention, norms/RoPE/layout kernels) and the whole HIP DiT forward against the
CPU oracle (oracle/dit.py, pinned to the reference's in-tree WanModel) and the twin's golden outputs.

Tolerances (stated, bf16 tensor-core path vs the fp32 oracle on identical bf16-rounded inputs):
  GEMM / attention: bf16 output rounding (2^-8 relative) + fp32 accumulation order -> |err| <= 1e-2 * max|ref|
  whole model: relative L2 error <= 2e-2 (bf16 activations between layers, as the reference's autocast path)."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import dit as odit

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF, F32 = torch.bfloat16, torch.float32


def _rand(shape, seed, scale=1.0):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed)) * scale


def _rel(got, want):
    return (got.float().cpu() - want.float()).abs().max().item() / (want.float().abs().max().item() + 1e-12)


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (257, 384, 144), (1, 512, 256), (300, 132, 200), (1000, 64, 1280),
                                   (2050, 1100, 520),
                                   # large shapes with K % 64 == 0 take the 256x256 ping-pong kernel (ragged M and N included)
                                   (2048, 2048, 512), (4100, 1028, 640), (16384, 256, 64), (3000, 5120, 1280)])
@pytest.mark.parametrize("epi", [0, 1, 2, 3, 4])
def test_gemm(M, N, K, epi):
    from worldforge_amd import dit
    x = _rand((M, K), 1).to(BF)
    w = (_rand((N, K), 2) / math.sqrt(K)).to(BF)
    b = _rand((N,), 3, 0.1)
    gate = _rand((N,), 4)
    ref = x.float() @ w.float().t() + b
    old = _rand((M, N), 5)
    if epi == 1:
        ref = torch.nn.functional.gelu(ref, approximate="tanh")
    if epi == 3:
        ref = old + ref * gate
    if epi == 4:
        ref = old + ref
    out_dtype = BF if epi in (0, 1) else F32
    out = old.to(DEV).clone() if epi in (3, 4) else torch.full((M, N), float("nan"), dtype=out_dtype, device=DEV)
    dit.gemm(x.to(DEV), w.to(DEV), b.to(DEV), out, epi, gate=gate.to(DEV) if epi == 3 else None)
    assert torch.isfinite(out).all()
    tol = 1e-2 if epi in (0, 1) else 2e-5 * math.sqrt(K) + 1e-5
    assert _rel(out, ref) <= tol, (_rel(out, ref), tol)


@pytest.mark.parametrize("M,N,K", [(4095, 5120, 640), (8190, 5120, 128), (4096, 640, 192 * 64)])
@pytest.mark.parametrize("epi", [0, 1, 3])
def test_gemm_wide_tile_equals_small_tile_kernel(M, N, K, epi):
    """Shapes whose last round of workgroups is fuller with 320-feature tiles (N % 320 == 0) take the 256 x 320 variant of the
    ping-pong kernel.  Every output element is the same sequence of MFMA accumulations over K in all tile shapes, so the result
    must equal, bit for bit, the same product computed in row chunks small enough for the 128 x 128 kernel."""
    from worldforge_amd import dit
    x = _rand((M, K), 21).to(BF).to(DEV)
    w = (_rand((N, K), 22) / math.sqrt(K)).to(BF).to(DEV)
    b = _rand((N,), 23, 0.1).to(DEV)
    gate = _rand((N,), 24).to(DEV) if epi == 3 else None
    old = _rand((M, N), 25).to(DEV)
    dt = BF if epi in (0, 1) else F32
    full = old.clone() if epi == 3 else torch.full((M, N), float("nan"), dtype=dt, device=DEV)
    dit.gemm(x, w, b, full, epi, gate=gate)
    parts = old.clone() if epi == 3 else torch.full((M, N), float("nan"), dtype=dt, device=DEV)
    for r0 in range(0, M, 1000):
        dit.gemm(x[r0:r0 + 1000], w, b, parts[r0:r0 + 1000], epi, gate=gate)
    assert torch.isfinite(full.float()).all()
    assert torch.equal(full, parts), (full.float() - parts.float()).abs().max()
    ref = x.float() @ w.float().t() + b
    if epi == 1:
        ref = torch.nn.functional.gelu(ref, approximate="tanh")
    if epi == 3:
        ref = old + ref * gate
    assert _rel(full, ref.cpu()) <= (1e-2 if epi in (0, 1) else 2e-5 * math.sqrt(K) + 1e-5)


@pytest.mark.parametrize("name,f16", [("wf_gemm_bf16_batched", False), ("wf_gemm_f16_batched", True)])
@pytest.mark.parametrize("B,M,N,K,epi", [(8, 1560, 384, 1152, 2),    # large together: the ping-pong kernel with the batch index on gridDim.y (ragged M / N tiles)
                                         (16, 1100, 264, 128, 0),    # the same with a 16-bit output
                                         (4, 300, 72, 136, 2)])      # small problems (K not a multiple of 64): the 128 x 128 kernel
def test_gemm_batched_equals_the_per_problem_calls(name, f16, B, M, N, K, epi):
    """`batch` independent products in one launch (the VAE mid-block's P . V of all frames; the block scores of the sparse attention): whichever
    kernel the batched call takes, problem b must equal -- bit for bit -- the single call on its operands, which takes the 128 x 128 kernel or the
    ping-pong kernel by its OWN size (every output element is the same K-ordered MFMA accumulation in both)."""
    from worldforge_amd import ops
    from worldforge_amd._ffi import call
    OP = torch.float16 if f16 else BF
    x = _rand((B, M, K), 31).to(OP).to(DEV)
    w = (_rand((B, N, K), 32) / math.sqrt(K)).to(OP).to(DEV)
    dt = OP if epi == 0 else F32
    out = torch.full((B, M, N), float("nan"), dtype=dt, device=DEV)
    call(name, x.data_ptr(), w.data_ptr(), out.data_ptr(), B, M, N, K, K, K, N, M * K, N * K, M * N, epi, ops.stream())
    single = "wf_gemm_f16" if f16 else "wf_gemm_bf16"
    for b in range(B):
        one = torch.full((M, N), float("nan"), dtype=dt, device=DEV)
        if f16:
            call(single, x[b].data_ptr(), w[b].data_ptr(), None, one.data_ptr(), M, N, K, K, K, N, epi, 1.0, ops.stream())
        else:
            call(single, x[b].data_ptr(), w[b].data_ptr(), None, one.data_ptr(), None, M, N, K, K, K, N, epi, ops.stream())
        assert torch.isfinite(one.float()).all()
        assert torch.equal(out[b], one), (b, (out[b].float() - one.float()).abs().max())
    ref = torch.einsum("bmk,bnk->bmn", x.float().cpu(), w.float().cpu())
    assert _rel(out.cpu(), ref) <= (1e-2 if epi == 0 else 2e-5 * math.sqrt(K) + 1e-5)


def test_gemm_strided_views_and_no_bias():
    from worldforge_amd import dit
    big = _rand((200, 512), 6).to(BF).to(DEV)
    x = big[:, 128:128 + 256]                      # ldx = 512
    w = (_rand((96 * 4, 256), 7) / 16).to(BF)
    outbig = torch.zeros((200, 1024), dtype=F32, device=DEV)
    out = outbig[:, 512:512 + 384]                 # ldo = 1024
    dit.gemm(x, w.to(DEV), None, out, 2)
    ref = x.float().cpu() @ w.float().t()
    assert _rel(out, ref) <= 1e-4
    assert outbig[:, :512].abs().max().item() == 0 and outbig[:, 896:].abs().max().item() == 0


def _attn_ref(q, k, v, scale):
    # q [H,Lq,128], k/v [H,Lk,128] fp32 -> [Lq, H*128]
    s = torch.einsum("hqd,hkd->hqk", q, k) * scale
    p = torch.softmax(s, dim=-1)
    o = torch.einsum("hqk,hkd->hqd", p, v)
    return o.permute(1, 0, 2).reshape(q.shape[1], -1)


def _to_layouts(q, k, v):
    """q,k,v [L, H*128] bf16 (token-major) -> device attention layouts via the product kernels."""
    from worldforge_amd import _ffi, ops
    H = q.shape[1] // 128
    Lq, Lk = q.shape[0], k.shape[0]
    Lkp = (Lk + 63) // 64 * 64
    ones = torch.ones(H * 128, device=DEV)
    qh = torch.empty((H, Lq, 128), dtype=BF, device=DEV)
    kh = torch.zeros((H, Lkp, 128), dtype=BF, device=DEV)
    vt = torch.empty((H, Lkp // 64, 128, 64), dtype=BF, device=DEV)
    return H, Lq, Lk, Lkp, qh, kh, vt


@pytest.mark.parametrize("H,Lq,Lk", [(2, 256, 64), (1, 300, 257), (3, 77, 512), (2, 1000, 1000), (1, 4524, 4524)])
def test_attention(H, Lq, Lk):
    from worldforge_amd import dit
    scale = 1 / math.sqrt(128)
    q = _rand((H, Lq, 128), 10).to(BF)
    k = _rand((H, Lk, 128), 11).to(BF)
    v = _rand((H, Lk, 128), 12).to(BF)
    Lkp = (Lk + 63) // 64 * 64
    kp = torch.zeros((H, Lkp, 128), dtype=BF)
    kp[:, :Lk] = k
    vp = torch.zeros((H, Lkp, 128), dtype=BF)
    vp[:, :Lk] = v
    vt = vp.view(H, Lkp // 64, 64, 128).transpose(2, 3).contiguous()
    out = torch.full((Lq, H * 128), float("nan"), dtype=BF, device=DEV)
    dit.attention(q.to(DEV), kp.to(DEV), vt.to(DEV), out, Lk, scale)
    ref = _attn_ref(q.float(), k.float(), v.float(), scale)
    assert torch.isfinite(out).all()
    assert _rel(out, ref) <= 1e-2, _rel(out, ref)
    # accumulate mode (second cross-attention, model.py:227)
    dit.attention(q.to(DEV), kp.to(DEV), vt.to(DEV), out, Lk, scale, accumulate=True)
    assert _rel(out, 2 * ref) <= 1.5e-2


@pytest.mark.parametrize("H,Lq,n1,n2", [(40, 1000, 257, 512),      # the DiT's shapes: 257 CLIP tokens (5 tiles, 1 valid key in the last), 512 text rows
                                        (2, 300, 40, 64),          # context 1 is a single ragged tile (the prologue masks it)
                                        (3, 77, 128, 500),         # ragged context 2, seam after an even number of tiles
                                        (1, 4524, 65, 129),        # both ragged, odd tile counts
                                        (2, 513, 320, 64)])
def test_fused_two_context_cross_attention_is_bit_identical_to_two_launches(H, Lq, n1, n2):
    """wf_attn_cross2_fwd (model.py:202-229 in one launch) == wf_attn_fwd(context 1) + wf_attn_fwd(context 2, accumulate), bit for bit, and
    both match the fp32 statement softmax(q k1^T) v1 + softmax(q k2^T) v2."""
    from worldforge_amd import dit
    scale = 1 / math.sqrt(128)
    pad = lambda n: (n + 63) // 64 * 64  # noqa: E731
    q = _rand((H, Lq, 128), 20).to(BF)
    ks = [_rand((H, n, 128), 21 + i).to(BF) for i, n in enumerate((n1, n2))]
    vs = [_rand((H, n, 128), 23 + i).to(BF) for i, n in enumerate((n1, n2))]
    kp, vt = [], []
    for k, v, n in zip(ks, vs, (n1, n2)):
        kk = torch.zeros((H, pad(n), 128), dtype=BF)
        kk[:, :n] = k
        vv = torch.zeros((H, pad(n), 128), dtype=BF)
        vv[:, :n] = v
        kp.append(kk.to(DEV))
        vt.append(vv.view(H, pad(n) // 64, 64, 128).transpose(2, 3).contiguous().to(DEV))
    two = torch.full((Lq, H * 128), float("nan"), dtype=BF, device=DEV)
    dit.attention(q.to(DEV), kp[0], vt[0], two, n1, scale)
    dit.attention(q.to(DEV), kp[1], vt[1], two, n2, scale, accumulate=True)
    kc = torch.cat(kp, dim=1).contiguous()
    vtc = torch.cat(vt, dim=1).contiguous()
    one = torch.full((Lq, H * 128), float("nan"), dtype=BF, device=DEV)
    dit.cross_attention2(q.to(DEV), kc, vtc, one, pad(n1), n1, n2, scale)
    assert torch.isfinite(one).all()
    assert torch.equal(one, two), (one.float() - two.float()).abs().max()
    ref = _attn_ref(q.float(), ks[0].float(), vs[0].float(), scale) + _attn_ref(q.float(), ks[1].float(), vs[1].float(), scale)
    assert _rel(one, ref) <= 1.5e-2, _rel(one, ref)
    # garbage in the padded key rows of either context must not leak
    kc2 = kc.clone()
    if pad(n1) > n1:
        kc2[:, n1:pad(n1)] = 1e4
    if pad(n2) > n2:
        kc2[:, pad(n1) + n2:] = -1e4
    again = torch.empty_like(one)
    dit.cross_attention2(q.to(DEV), kc2, vtc, again, pad(n1), n1, n2, scale)
    assert torch.equal(again, one)


@pytest.mark.parametrize("H,Lq,Lk,nsplit,segs", [(2, 300, 1000, 2, 1), (1, 256, 4524, 3, 1), (3, 77, 640, 2, 1), (2, 500, 1024, 2, 4),
                                                  (1, 128, 8192 + 37, 8, 1)])
def test_attention_kv_splits(H, Lq, Lk, nsplit, segs):
    """wf_attn_fwd_split: KV sweep split nsplit ways + exact merge (used when one rank's token shard is too short to fill the chip).
    Ragged last tile, spiked keys in the LAST split (its reference max differs from the others'), all-gathered K/V segments."""
    from worldforge_amd import dit
    scale = 1 / math.sqrt(128)
    q = _rand((H, Lq, 128), 30).to(BF)
    k = _rand((H, Lk, 128), 31)
    k[:, Lk - 3] = q[:, 5].float() * 4.0     # a late spike: the last split sees a much larger row max for query 5
    k = k.to(BF)
    v = _rand((H, Lk, 128), 32).to(BF)
    Lkp = (Lk + 63) // 64 * 64
    if segs > 1:
        assert Lkp % (segs * 64) == 0
    kp = torch.zeros((H, Lkp, 128), dtype=BF)
    kp[:, :Lk] = k
    vp = torch.zeros((H, Lkp, 128), dtype=BF)
    vp[:, :Lk] = v
    vt = vp.view(H, Lkp // 64, 64, 128).transpose(2, 3).contiguous()
    kd, vd = kp.to(DEV), vt.to(DEV)
    if segs > 1:  # shard-major layout of the sequence-parallel DiT: [P][H][seg][128], [P][H][seg/64][128][64]
        seg = Lkp // segs
        kd = kd.view(H, segs, seg, 128).transpose(0, 1).contiguous()
        vd = vd.view(H, segs, seg // 64, 128, 64).transpose(0, 1).contiguous()
    ref = _attn_ref(q.float(), k.float(), v.float(), scale)
    one = torch.empty((Lq, H * 128), dtype=BF, device=DEV)
    dit.attention(q.to(DEV), kd, vd, one, Lk, scale, nsplit=1)
    out = torch.full((Lq, H * 128), float("nan"), dtype=BF, device=DEV)
    dit.attention(q.to(DEV), kd, vd, out, Lk, scale, nsplit=nsplit)
    assert torch.isfinite(out).all()
    assert _rel(out, ref) <= 1e-2, _rel(out, ref)
    assert _rel(out, one.float().cpu()) <= 6e-3        # same math, different association: bf16 output rounding apart
    dit.attention(q.to(DEV), kd, vd, out, Lk, scale, accumulate=True, nsplit=nsplit)
    assert _rel(out, 2 * ref) <= 1.5e-2


def test_attention_online_softmax_rescale_is_exercised():
    """Spike late keys so the running maximum jumps in the last tiles (forces the O / l rescale branch)."""
    from worldforge_amd import dit
    H, L = 1, 512
    scale = 1 / math.sqrt(128)
    q = _rand((H, L, 128), 13).to(BF)
    k = _rand((H, L, 128), 14)
    k[:, 300] = q[0, 5].float() * 3.0
    k[:, 500] = q[0, 17].float() * 5.0
    k = k.to(BF)
    v = _rand((H, L, 128), 15).to(BF)
    vt = v.view(H, L // 64, 64, 128).transpose(2, 3).contiguous()
    out = torch.empty((L, 128), dtype=BF, device=DEV)
    dit.attention(q.to(DEV), k.to(DEV), vt.to(DEV), out, L, scale)
    ref = _attn_ref(q.float(), k.float(), v.float(), scale)
    assert _rel(out, ref) <= 1e-2


@pytest.mark.parametrize("H,Lq,Lk,nsplit", [(2, 300, 1000, 1), (1, 256, 4524, 1), (3, 77, 640, 1), (1, 128, 8192 + 37, 1), (2, 300, 1000, 2),
                                            (1, 128, 8192 + 37, 8), (2, 64, 40, 1)])
def test_attention_prescaled_q(H, Lq, Lk, nsplit):
    """softmax_scale = 0 (k_attn_w4<4>): Q arrives multiplied by head_dim^-1/2 * log2(e), the score accumulators start from -m.  The
    oracle gets the very same bf16 Q values divided by that factor, so the two compute the same function; late key spikes force the
    rescale branch (running max grows by far more than 2^8) and re-base the already-computed scores of the next tile."""
    from worldforge_amd import dit
    scale = 1 / math.sqrt(128)
    alpha = scale * 1.4426950408889634
    q0 = _rand((H, Lq, 128), 40)
    k = _rand((H, Lk, 128), 41)
    if Lk > 600:
        k[:, 300] = q0[:, 5] * 3.0
        k[:, Lk - 70] = q0[:, 17 % Lq] * 5.0
        k[:, Lk - 3] = q0[:, 9] * 7.0
    k = k.to(BF)
    qs = (q0 * alpha).to(BF)                       # what wf_rmsnorm_heads(out_scale = alpha) hands over
    v = _rand((H, Lk, 128), 42).to(BF)
    Lkp = (Lk + 63) // 64 * 64
    kp = torch.zeros((H, Lkp, 128), dtype=BF)
    kp[:, :Lk] = k
    vp = torch.zeros((H, Lkp, 128), dtype=BF)
    vp[:, :Lk] = v
    vt = vp.view(H, Lkp // 64, 64, 128).transpose(2, 3).contiguous()
    out = torch.full((Lq, H * 128), float("nan"), dtype=BF, device=DEV)
    dit.attention(qs.to(DEV), kp.to(DEV), vt.to(DEV), out, Lk, 0.0, nsplit=nsplit)
    ref = _attn_ref(qs.float() / alpha, k.float(), v.float(), scale)
    assert torch.isfinite(out).all()
    assert _rel(out, ref) <= 1e-2, _rel(out, ref)
    # the in-kernel-scale path on the same operands is the same function (bf16 output rounding apart)
    qf = (qs.float() / alpha)
    if torch.equal(qf.to(BF).float(), qf):   # only comparable when the un-scaled Q is representable (it is not in general)
        two = torch.empty_like(out)
        dit.attention(qf.to(BF).to(DEV), kp.to(DEV), vt.to(DEV), two, Lk, scale, nsplit=nsplit)
        assert _rel(out, two.float().cpu()) <= 6e-3
    dit.attention(qs.to(DEV), kp.to(DEV), vt.to(DEV), out, Lk, 0.0, accumulate=True, nsplit=nsplit)
    assert _rel(out, 2 * ref) <= 1.5e-2


@pytest.mark.parametrize("H,Lq,Lk,nsplit,kscale", [(2, 300, 1000, 1, 1.0), (1, 256, 4524, 1, 1.0), (1, 128, 8192 + 37, 8, 1.0), (2, 300, 1000, 1, 40.0),
                                                   (2, 64, 40, 1, 1.0)])
def test_attention_prescaled_q_with_key_norm_bound(H, Lq, Lk, nsplit, kscale):
    """kmax2 / qmax2 given: heads whose Cauchy-Schwarz bound B = max|q| max|k| is <= 50 (exp2 domain) run without running-max tracking --
    the same function, within tolerance; keys scaled by 40 push B past 50 so that the tracked fallback runs (late spikes included in both;
    with the unit-scale spikes B is ~ 30-45 here, so the un-tracked path does see scores far above its first-tile reference max)."""
    from worldforge_amd import dit
    scale = 1 / math.sqrt(128)
    alpha = scale * 1.4426950408889634
    q0 = _rand((H, Lq, 128), 60)
    k = _rand((H, Lk, 128), 61) * kscale
    if Lk > 600:
        k[:, 300] = q0[:, 5] * 3.0
        k[:, Lk - 3] = q0[:, 9] * 7.0
    k = k.to(BF)
    qs = (q0 * alpha).to(BF)
    v = _rand((H, Lk, 128), 62).to(BF)
    Lkp = (Lk + 63) // 64 * 64
    kp = torch.zeros((H, Lkp, 128), dtype=BF)
    kp[:, :Lk] = k
    vp = torch.zeros((H, Lkp, 128), dtype=BF)
    vp[:, :Lk] = v
    vt = vp.view(H, Lkp // 64, 64, 128).transpose(2, 3).contiguous()
    kd = kp.to(DEV)
    km = dit.head_max_norm2(kd, Lk, torch.empty(H, device=DEV))
    qm = dit.head_max_norm2(qs.to(DEV), Lq, torch.empty(H, device=DEV))
    want_km = (k.float() ** 2).sum(-1).max(dim=1).values
    assert torch.allclose(km.cpu(), want_km, rtol=1e-5)
    assert torch.allclose(qm.cpu(), (qs.float() ** 2).sum(-1).max(dim=1).values, rtol=1e-5)
    print("B per head", (km * qm).sqrt().cpu().tolist())
    out = torch.full((Lq, H * 128), float("nan"), dtype=BF, device=DEV)
    dit.attention(qs.to(DEV), kd, vt.to(DEV), out, Lk, 0.0, nsplit=nsplit, kmax2=km, qmax2=qm)
    ref = _attn_ref(qs.float() / alpha, k.float(), v.float(), scale)
    assert torch.isfinite(out).all()
    assert _rel(out, ref) <= 1e-2, _rel(out, ref)
    trk = torch.empty_like(out)
    dit.attention(qs.to(DEV), kd, vt.to(DEV), trk, Lk, 0.0, nsplit=nsplit)
    assert _rel(out, trk.float().cpu()) <= 6e-3
    # two shard vectors (all-gathered K): the kernel takes the max over them
    km2 = torch.stack([km * 0.25, km]).contiguous()
    out2 = torch.empty_like(out)
    dit.attention(qs.to(DEV), kd, vt.to(DEV), out2, Lk, 0.0, nsplit=nsplit, kmax2=km2, qmax2=qm)
    assert torch.equal(out2, out)


def test_rmsnorm_heads_out_scale_is_applied_before_the_rounding():
    from worldforge_amd import _ffi, dit, ops
    L, H, f, h, w = 40, 2, 2, 4, 5
    C = H * 128
    x = _rand((L, C), 50).to(BF).to(DEV)
    wq = (1 + 0.1 * _rand((C,), 51)).to(DEV)
    cos, sin = dit.rope_tables(128, f, h, w)
    cd, sd = cos.to(DEV), sin.to(DEV)
    a, b = torch.zeros((H, L, 128), dtype=BF, device=DEV), torch.zeros((H, L, 128), dtype=BF, device=DEV)
    alpha = 1.4426950408889634 / math.sqrt(128)
    _ffi.call("wf_rmsnorm_heads", x.data_ptr(), C, wq.data_ptr(), cd.data_ptr(), sd.data_ptr(), a.data_ptr(), L, L, C, 1e-6, 1.0, ops.stream())
    _ffi.call("wf_rmsnorm_heads", x.data_ptr(), C, wq.data_ptr(), cd.data_ptr(), sd.data_ptr(), b.data_ptr(), L, L, C, 1e-6, alpha, ops.stream())
    # b = bf16(y * alpha) with y the un-rounded result: within half a bf16 ulp of alpha * y, i.e. 2^-8 relative of bf16(y) * alpha
    assert ((b.float() - a.float() * alpha).abs() <= 2.0 ** -7 * (a.float() * alpha).abs() + 1e-8).all()
    assert not torch.equal(b.float(), (a.float() * alpha).to(BF).float())   # not the double rounding


@pytest.mark.parametrize("L,C,rope", [(300, 5120, True), (77, 256, False), (1030, 1024, True)])
def test_rmsnorm_heads_bound_equals_the_separate_norm_pass(L, C, rope):
    """wf_rmsnorm_heads_bound: same output as wf_rmsnorm_heads, and max |row|^2 per head equal to wf_head_max_norm2 of that output (the values
    as stored; fp32 sums in a different order); a NaN row reports +inf for its head only."""
    from worldforge_amd import dit
    H = C // 128
    m = dit.WanTransformer3DModel(dit.DiTConfig(dim=C, ffn_dim=2 * C, num_heads=H, num_layers=1, text_dim=64), DEV)
    src = _rand((L, 3 * C), 5).to(BF).to(DEV)
    w = (1 + 0.05 * _rand((C,), 6)).to(DEV)
    cos, sin = (m._rope_tables(1, 1, L) if rope else (None, None))
    Lp = (L + 63) // 64 * 64
    a = torch.zeros((H, Lp, 128), dtype=BF, device=DEV)
    b = torch.zeros((H, Lp, 128), dtype=BF, device=DEV)
    bound = torch.full((H,), -1.0, dtype=F32, device=DEV)
    m._heads(src, C, w, cos, sin, a, L, out_scale=0.1275)
    m._heads(src, C, w, cos, sin, b, L, out_scale=0.1275, bound=bound)
    assert torch.equal(a, b)
    want = dit.head_max_norm2(a, L, torch.empty(H, dtype=F32, device=DEV))
    assert torch.allclose(bound, want, rtol=1e-5, atol=0), (bound - want).abs().max()
    assert float(bound.min()) > 0
    src[L // 2, C + 128 * (H - 1) + 3] = float("nan")      # one element of the last head's input... the RMS norm spreads it over the whole row
    m._heads(src, C, w, cos, sin, b, L, out_scale=0.1275, bound=bound)
    assert torch.isinf(bound).all()                         # (RMS over all channels: every head of that row is NaN -> every head's bound is +inf)


def test_ln_modulate_and_affine():
    from worldforge_amd import _ffi, ops
    for C in (256, 1280, 5120):
        x = _rand((37, C), 20, 3.0) + 0.5
        mul, add = _rand((C,), 21, 0.3), _rand((C,), 22, 0.3)
        xd, md, ad = x.to(DEV), mul.to(DEV), add.to(DEV)  # keep device tensors alive across the raw-pointer call
        for plus_one, out_dt in ((1, BF), (0, F32)):
            out = torch.empty((37, C), dtype=out_dt, device=DEV)
            _ffi.call("wf_ln_modulate", xd.data_ptr(), md.data_ptr(), ad.data_ptr(), out.data_ptr(),
                      1 if out_dt == BF else 0, 37, C, 1e-6, plus_one, ops.stream())
            ref = odit.layer_norm(x, 1e-6) * (plus_one + mul) + add
            tol = 8e-3 if out_dt == BF else 1e-5
            assert (out.float().cpu() - ref).abs().max().item() <= tol * ref.abs().max().item()


def test_rmsnorm_rope_heads_and_v_transpose():
    from worldforge_amd import _ffi, dit, ops
    f, h, w = 2, 3, 5
    L, H = f * h * w, 3
    C = H * 128
    qkv = _rand((L, 3 * C), 30).to(BF)
    wq = 1 + 0.1 * _rand((C,), 31)
    cos, sin = dit.rope_tables(128, f, h, w)
    out = torch.zeros((H, 64, 128), dtype=BF, device=DEV)
    qd, wqd, cd, sd = qkv.to(DEV), wq.to(DEV), cos.to(DEV), sin.to(DEV)
    _ffi.call("wf_rmsnorm_heads", qd[:, C:2 * C].data_ptr(), 3 * C, wqd.data_ptr(), cd.data_ptr(),
              sd.data_ptr(), out.data_ptr(), L, 64, C, 1e-6, 1.0, ops.stream())
    kk = qkv[:, C:2 * C].float()
    ref = odit.rope_apply(odit.rms_norm(kk, wq, 1e-6).view(L, H, 128), odit.rope_tables(128, f, h, w))
    got = out[:, :L].permute(1, 0, 2).float().cpu()
    assert (got - ref).abs().max().item() <= 2e-2 * ref.abs().max().item()
    assert out[:, L:].abs().max().item() == 0
    # no-rope variant
    _ffi.call("wf_rmsnorm_heads", qd.data_ptr(), 3 * C, wqd.data_ptr(), None, None, out.data_ptr(), L, 64, C, 1e-6, 1.0,
              ops.stream())
    ref2 = odit.rms_norm(qkv[:, :C].float(), wq, 1e-6).view(L, H, 128)
    assert (out[:, :L].permute(1, 0, 2).float().cpu() - ref2).abs().max().item() <= 2e-2 * ref2.abs().max().item()
    # V transpose
    vt = torch.full((H, 1, 128, 64), float("nan"), dtype=BF, device=DEV)
    _ffi.call("wf_v_transpose", qd[:, 2 * C:].data_ptr(), 3 * C, vt.data_ptr(), L, 64, H, ops.stream())
    v = qkv[:, 2 * C:].view(L, H, 128)
    want = torch.zeros((H, 64, 128), dtype=BF)
    want[:, :L] = v.permute(1, 0, 2)
    assert torch.equal(vt.cpu()[:, 0], want.transpose(1, 2))


def test_patchify_unpatchify():
    from worldforge_amd import _ffi, ops
    Cin, T, Hh, Ww = 36, 2, 6, 8
    x = _rand((Cin, T, Hh, Ww), 40).to(BF)
    tok = torch.empty((T * 3 * 4, Cin * 4), dtype=BF, device=DEV)
    xd = x.to(DEV)
    _ffi.call("wf_patchify", xd.data_ptr(), tok.data_ptr(), Cin, T, Hh, Ww, ops.stream())
    want = x.view(Cin, T, 3, 2, 4, 2).permute(1, 2, 4, 0, 3, 5).reshape(T * 12, Cin * 4)
    assert torch.equal(tok.cpu(), want)
    y = _rand((T * 12, 64), 41)
    out = torch.empty((16, T, Hh, Ww), dtype=F32, device=DEV)
    yd = y.to(DEV)
    _ffi.call("wf_unpatchify", yd.data_ptr(), out.data_ptr(), 16, T, Hh, Ww, ops.stream())
    u = torch.einsum("fhwpqrc->cfphqwr", y.view(T, 3, 4, 1, 2, 2, 16)).reshape(16, T, Hh, Ww)
    assert torch.equal(out.cpu(), u)


@pytest.mark.parametrize("name", ["tiny", "odd"])
def test_dit_forward_vs_twin_golden_and_oracle(name, golden_dir):
    from worldforge_amd import dit
    g = np.load(os.path.join(golden_dir, "g7_dit.npz"))
    dim, heads, ffn, layers, T, h, w = g[f"{name}_cfg"].tolist()
    ocfg = odit.DiTConfig(dim=dim, ffn_dim=ffn, num_heads=heads, num_layers=layers, text_dim=64)
    W = odit.random_weights(ocfg, seed=11)
    cfg = dit.DiTConfig(dim=dim, ffn_dim=ffn, num_heads=heads, num_layers=layers, text_dim=64)
    model = dit.WanTransformer3DModel(cfg, DEV).load_state_dict(W)
    x = torch.from_numpy(g[f"{name}_x"])
    ctx, clip = torch.from_numpy(g[f"{name}_ctx"]), torch.from_numpy(g[f"{name}_clip"])
    out = model.forward_tokens(x.to(BF).to(DEV), 749.0, ctx.to(BF).to(DEV), clip.to(BF).to(DEV)).cpu()
    twin = torch.from_numpy(g[f"{name}_out"])
    # oracle on the same bf16-rounded weights / inputs isolates kernel error from input rounding
    Wb = {k: (v.to(BF).float() if v.dim() >= 2 else v) for k, v in W.items()}
    orc = odit.forward(Wb, ocfg, x.to(BF).float(), torch.tensor(749), ctx.to(BF).float(), clip.to(BF).float())
    e_orc = (out - orc).norm().item() / orc.norm().item()
    e_twin = (out - twin).norm().item() / twin.norm().item()
    print(f"[{name}] rel L2 vs oracle(bf16 weights) {e_orc:.3e}, vs twin golden (fp32 weights) {e_twin:.3e}")
    assert e_orc <= 2e-2 and e_twin <= 3e-2


def test_dit_call_protocol_matches_diffusers_signature():
    from worldforge_amd import dit
    cfg = dit.DiTConfig(dim=256, ffn_dim=512, num_heads=2, num_layers=1, text_dim=64)
    m = dit.WanTransformer3DModel(cfg, DEV).init_random(0)
    x = _rand((1, 36, 2, 8, 8), 50).to(BF).to(DEV)
    out = m(hidden_states=x, timestep=torch.tensor([499]), encoder_hidden_states=_rand((1, 20, 64), 51).to(BF).to(DEV),
            encoder_hidden_states_image=_rand((1, 257, 1280), 52).to(BF).to(DEV), attention_kwargs=None, return_dict=False)[0]
    assert out.shape == (1, 16, 2, 8, 8) and out.dtype == BF and torch.isfinite(out).all()
    with pytest.raises(NotImplementedError):
        m(hidden_states=torch.cat([x, x]), timestep=torch.tensor([1]), encoder_hidden_states=None)


def test_sequence_parallel_path_world1_matches_single():
    """The N > 1 code path (token shard plan, K / V^T all-gather over RCCL, segment-addressed attention, velocity
    gather) exercised with a 1-rank RCCL group on the single test GPU: must equal the single-GPU path bit for bit."""
    import os
    from worldforge_amd import dit, parallel
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29577")
    comm = parallel.init(1, 0, 0, backend="nccl")
    cfg = dit.DiTConfig(dim=256, ffn_dim=512, num_heads=2, num_layers=2, text_dim=64)
    x = _rand((36, 3, 8, 10), 60).to(BF).to(DEV)
    ctx, clip = _rand((30, 64), 61).to(BF).to(DEV), _rand((257, 1280), 62).to(BF).to(DEV)
    m0 = dit.WanTransformer3DModel(cfg, DEV).init_random(5)
    ref = m0.forward_tokens(x, 500.0, ctx, clip).clone()
    m1 = dit.WanTransformer3DModel(cfg, DEV, comm=comm)
    m1.w = m0.w
    got = m1.forward_tokens(x, 500.0, ctx, clip)
    assert torch.equal(got, ref)
    # CFG pair in lock-step: the exchanges run on the communication stream under the other branch's layer (real streams / events)
    ctx_b = _rand((17, 64), 63).to(BF).to(DEV)
    ref_b = m0.forward_tokens(x, 500.0, ctx_b, clip).clone()
    for _ in range(3):
        a, b = m1.forward_tokens_pair(x, 500.0, ctx, ctx_b, clip)
        assert torch.equal(a, ref) and torch.equal(b, ref_b)
    torch.distributed.destroy_process_group()


def test_context_kv_cache_is_transparent(monkeypatch):
    """The per-layer cross-attention K / V of a (text, image) context are cached across forwards (on by default since round 5;
    WF_CTX_CACHE=0 = the reference's recompute-every-forward, this test's reference arm): results must equal the uncached forward bit
    for bit, a second context must not hit the first one's entry, and an in-place edit of the embeddings (version counter) must
    invalidate the entry."""
    from worldforge_amd import dit
    cfg = dit.DiTConfig(dim=256, ffn_dim=512, num_heads=2, num_layers=2, text_dim=64)
    x = _rand((36, 3, 8, 10), 80).to(BF).to(DEV)
    ca, cb = _rand((30, 64), 81).to(BF).to(DEV), _rand((30, 64), 82).to(BF).to(DEV)
    clip = _rand((257, 1280), 83).to(BF).to(DEV)
    m = dit.WanTransformer3DModel(cfg, DEV).init_random(5)
    monkeypatch.setenv("WF_CTX_CACHE", "0")
    ref_a = m.forward_tokens(x, 500.0, ca, clip).clone()
    ref_b = m.forward_tokens(x, 500.0, cb, clip).clone()
    assert not hasattr(m, "_ctx_cache")
    monkeypatch.delenv("WF_CTX_CACHE")   # the default
    for _ in range(2):   # second round: served from the cache
        assert torch.equal(m.forward_tokens(x, 500.0, ca, clip), ref_a)
        assert torch.equal(m.forward_tokens(x, 250.0, cb, clip), m.forward_tokens(x, 250.0, cb, clip))
        assert torch.equal(m.forward_tokens(x, 500.0, cb, clip), ref_b)
    assert len(m._ctx_cache) == 2
    ca.mul_(0.5)         # in-place edit: same storage, new version -> must not be served from the stale entry
    got_c = m.forward_tokens(x, 500.0, ca, clip).clone()
    monkeypatch.setenv("WF_CTX_CACHE", "0")
    ref_c = m.forward_tokens(x, 500.0, ca, clip)
    assert not torch.equal(ref_c, ref_a)
    assert torch.equal(got_c, ref_c)
    # weights: a reload (new dict) and a DECLARED in-place edit both start a new weights version -- never id() of a dict (the
    # address of a freed dict can be reused, and the prompt tensors the cache keeps alive would then hit K / V of the old weights)
    monkeypatch.delenv("WF_CTX_CACHE")
    before = m.forward_tokens(x, 500.0, cb, clip).clone()
    assert len(m._ctx_cache) >= 1
    v0 = m._wver
    m.init_random(6)
    assert m._wver == v0 + 1 and not hasattr(m, "_ctx_cache")
    after = m.forward_tokens(x, 500.0, cb, clip).clone()
    monkeypatch.setenv("WF_CTX_CACHE", "0")
    assert torch.equal(after, m.forward_tokens(x, 500.0, cb, clip)) and not torch.equal(after, before)
    monkeypatch.delenv("WF_CTX_CACHE")
    key = next(k for k in m.w if k.endswith("cross_attn.kv.w"))     # the prompt-context K / V projection: what the cache holds
    m.w[key].mul_(0.5)                    # e.g. a LoRA folded into the same tensor
    m.weights_changed()
    edited = m.forward_tokens(x, 500.0, cb, clip).clone()
    monkeypatch.setenv("WF_CTX_CACHE", "0")
    assert torch.equal(edited, m.forward_tokens(x, 500.0, cb, clip)) and not torch.equal(edited, after)


def test_cfg_pair_shares_the_prompt_independent_prefix_bit_identically():
    """forward_tokens_pair (PIPE:593-610: same latents and timestep, two prompts) computes the patch embedding and layer 0's self-attention
    block ONCE (round 6): both velocities must equal two independent forwards bit for bit, with one self-attention launch fewer."""
    from worldforge_amd import dit
    cfg = dit.DiTConfig(dim=256, ffn_dim=512, num_heads=2, num_layers=3, text_dim=64)
    x = _rand((36, 3, 8, 10), 90).to(BF).to(DEV)
    ca, cb = _rand((30, 64), 91).to(BF).to(DEV), _rand((12, 64), 92).to(BF).to(DEV)
    clip = _rand((257, 1280), 93).to(BF).to(DEV)
    m = dit.WanTransformer3DModel(cfg, DEV).init_random(5)
    ref_a = m.forward_tokens(x, 431.0, ca, clip).clone()
    ref_b = m.forward_tokens(x, 431.0, cb, clip).clone()
    assert not torch.equal(ref_a, ref_b)
    for share, launches in ((True, 2 * cfg.num_layers - 1), (False, 2 * cfg.num_layers)):
        m.pair_share_layer0 = share
        for _ in range(2):
            dit.PROFILE_ATTN = []
            a, b = m.forward_tokens_pair(x, 431.0, ca, cb, clip)
            n = len(dit.PROFILE_ATTN)
            dit.PROFILE_ATTN = None
            assert torch.equal(a, ref_a) and torch.equal(b, ref_b)
            assert n == launches, (share, n)
    # a different timestep / different latents in the next pair: nothing stale is reused
    m.pair_share_layer0 = True
    x2 = _rand((36, 3, 8, 10), 94).to(BF).to(DEV)
    a2, b2 = m.forward_tokens_pair(x2, 120.0, ca, cb, clip)
    assert torch.equal(a2, m.forward_tokens(x2, 120.0, ca, clip)) and torch.equal(b2, m.forward_tokens(x2, 120.0, cb, clip))


def test_diffusers_layout_transformer_checkpoint_loads(tmp_path):
    """A sharded diffusers-layout `transformer/` directory (config.json + 2 safetensors shards + index; names through the inverse of
    dit.diffusers_key_map, per-block / final scale_shift_table) -> WanTransformer3DModel.from_pretrained -> the same forward as the
    twin-keyed loader, bit for bit."""
    import json
    from safetensors.torch import save_file
    from worldforge_amd import dit
    cfg = dit.DiTConfig(dim=256, ffn_dim=512, num_heads=2, num_layers=2, text_dim=64)
    ocfg = odit.DiTConfig(dim=256, ffn_dim=512, num_heads=2, num_layers=2, text_dim=64)
    W = odit.random_weights(ocfg, seed=9)
    inv = {v: k for k, v in dit.diffusers_key_map(cfg.num_layers).items()}
    sd = {}
    for k, v in W.items():
        if k == "head.modulation":
            sd["scale_shift_table"] = v
        elif k.endswith(".modulation"):
            sd[k.replace(".modulation", ".scale_shift_table")] = v
        else:
            base, _, leaf = k.rpartition(".")
            sd[f"{inv[base]}.{leaf}"] = v
    sd = {k: v.to(BF).contiguous() for k, v in sd.items()}  # released checkpoints are bf16
    folder = tmp_path / "transformer"
    os.makedirs(folder)
    names = sorted(sd)
    half = len(names) // 2
    wm = {}
    for fn, ns in (("diffusion_pytorch_model-00001-of-00002.safetensors", names[:half]),
                   ("diffusion_pytorch_model-00002-of-00002.safetensors", names[half:])):
        save_file({n: sd[n] for n in ns}, str(folder / fn))
        wm.update({n: fn for n in ns})
    (folder / "diffusion_pytorch_model.safetensors.index.json").write_text(json.dumps({"metadata": {}, "weight_map": wm}))
    (folder / "config.json").write_text(json.dumps({"_class_name": "WanTransformer3DModel", "num_attention_heads": 2, "attention_head_dim": 128,
                                                     "ffn_dim": 512, "num_layers": 2, "in_channels": 36, "out_channels": 16, "text_dim": 64,
                                                     "freq_dim": 256, "image_dim": 1280, "patch_size": [1, 2, 2], "eps": 1e-6}))
    m1 = dit.WanTransformer3DModel.from_pretrained(str(tmp_path), device=DEV)
    m0 = dit.WanTransformer3DModel(cfg, DEV).load_state_dict({k: v.to(BF) for k, v in W.items()})
    assert m1.cfg == cfg
    x = _rand((36, 2, 8, 12), 60).to(BF).to(DEV)
    ctx, clip = _rand((30, 64), 61).to(BF).to(DEV), _rand((257, 1280), 62).to(BF).to(DEV)
    a, b = m0.forward_tokens(x, 500.0, ctx, clip).clone(), m1.forward_tokens(x, 500.0, ctx, clip).clone()
    assert torch.isfinite(a).all() and torch.equal(a, b)
