"""GPU: the HIP VAE (implicit-GEMM MFMA convolutions, whole-sequence causal processing) against goldens recorded from
the reference's chunked / cached WanVAE_ and against the CPU oracle.

Tolerance (stated): bf16 MFMA operands with fp32 accumulation and an fp32 residual stream vs the fp32 reference:
  relative L2 error <= 1.5e-2 on latents and pixels, max abs pixel error <= 6e-2 (pixels in [-1,1]); per-kernel checks
  are tighter."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import vae as ovae

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF, F32 = torch.bfloat16, torch.float32
CASES = ["f9_32x32", "f5_48x40", "f1_32x32", "f17_16x24"]


@pytest.fixture(scope="module")
def model():
    from worldforge_amd.vae import AutoencoderKLWan
    return AutoencoderKLWan(DEV, precision="bf16").load_state_dict(ovae.random_weights(seed=5))


def _rel(a, b):
    return (a - b).norm().item() / (b.norm().item() + 1e-12)


@pytest.mark.parametrize("name", CASES)
def test_encode_decode_vs_twin_goldens(name, model, golden_dir):
    g = np.load(os.path.join(golden_dir, "g8_vae.npz"))
    x, z = torch.from_numpy(g[f"{name}_x"]), torch.from_numpy(g[f"{name}_z"])
    mu = model.encode(x.to(DEV)).latent_dist.mode().cpu()
    dec = model.decode(z.to(DEV), return_dict=False)[0].cpu()
    mu_ref, dec_ref = torch.from_numpy(g[f"{name}_mu"]), torch.from_numpy(g[f"{name}_dec"])
    assert mu.shape == mu_ref.shape and dec.shape == dec_ref.shape
    e_mu, e_dec = _rel(mu, mu_ref), _rel(dec, dec_ref)
    print(f"[{name}] rel L2: mu {e_mu:.3e} dec {e_dec:.3e}; max abs: mu {(mu - mu_ref).abs().max():.3e} "
          f"dec {(dec - dec_ref).abs().max():.3e}")
    assert e_mu <= 1.5e-2 and e_dec <= 1.5e-2
    assert (dec - dec_ref).abs().max().item() <= 6e-2
    assert dec.abs().max().item() <= 1.0


def _conv_ref(x_cl, w, b, k, st, ss, pt, ps, up2=False):
    """x_cl [T,H,W,C] -> torch conv3d reference with the kernel's padding convention -> [T',H',W',Cout]."""
    x = x_cl.permute(3, 0, 1, 2).unsqueeze(0).float()
    if up2:
        x = F.interpolate(x, scale_factor=(1.0, 2.0, 2.0), mode="nearest-exact")
    pad_after_h = max(0, k[1] - 1 - ps) if ss == 1 else 1
    x = F.pad(x, (ps, pad_after_h, ps, pad_after_h, pt, 0))
    y = F.conv3d(x, w, b, stride=(st, ss, ss))
    return y[0].permute(1, 2, 3, 0)


@pytest.mark.parametrize("cfg", [
    dict(T=3, H=10, W=12, cin=96, cout=96, k=(3, 3, 3), st=1, ss=1, pt=2, ps=1),
    dict(T=2, H=9, W=7, cin=192, cout=384, k=(3, 3, 3), st=1, ss=1, pt=2, ps=1),
    dict(T=2, H=8, W=10, cin=96, cout=96, k=(1, 3, 3), st=1, ss=2, pt=0, ps=0),       # downsample conv (pad (0,1,0,1))
    dict(T=2, H=6, W=5, cin=192, cout=96, k=(1, 3, 3), st=1, ss=1, pt=0, ps=1, up2=True),  # upsample2x + conv
    dict(T=5, H=4, W=6, cin=384, cout=384, k=(3, 1, 1), st=2, ss=1, pt=0, ps=0),     # downsample3d time_conv
    dict(T=40, H=16, W=16, cin=96, cout=192, k=(3, 3, 3), st=1, ss=1, pt=2, ps=1),  # several pixel tiles, ragged M
    # >= 32768 pixels, stride 1: the 512-pixel DMA-gather ping-pong kernel (odd and even numbers of 32-channel K tiles, ragged M)
    dict(T=9, H=61, W=64, cin=96, cout=96, k=(3, 3, 3), st=1, ss=1, pt=2, ps=1),
    dict(T=5, H=80, W=90, cin=192, cout=192, k=(3, 3, 3), st=1, ss=1, pt=2, ps=1),
    dict(T=3, H=100, W=120, cin=384, cout=96, k=(1, 3, 3), st=1, ss=1, pt=0, ps=1),
    dict(T=33, H=32, W=32, cin=96, cout=384, k=(3, 1, 1), st=1, ss=1, pt=2, ps=0),
    dict(T=4, H=96, W=96, cin=32, cout=96, k=(3, 3, 3), st=1, ss=1, pt=2, ps=1),
])
def test_conv3d_cl_vs_torch(cfg):
    from worldforge_amd import _ffi, ops
    g = torch.Generator().manual_seed(3)
    T, H, W, cin, cout, k = cfg["T"], cfg["H"], cfg["W"], cfg["cin"], cfg["cout"], cfg["k"]
    st, ss, pt, ps, up2 = cfg["st"], cfg["ss"], cfg["pt"], cfg["ps"], cfg.get("up2", False)
    x = torch.randn(T, H, W, cin, generator=g).to(BF)
    w = (torch.randn(cout, cin, *k, generator=g) / math.sqrt(cin * math.prod(k))).to(BF)
    b = torch.randn(cout, generator=g) * 0.1
    ref = _conv_ref(x, w.float(), b, k, st, ss, pt, ps, up2)
    To, Ho, Wo, _ = ref.shape
    resid = torch.randn(To, Ho, Wo, cout, generator=g)
    wk = w.permute(0, 2, 3, 4, 1).reshape(cout, -1, cin).contiguous().to(DEV)
    xd, bd, rd = x.to(DEV), b.to(DEV), resid.to(DEV)
    zp = torch.zeros(64, dtype=BF, device=DEV)
    of = torch.full((To, Ho, Wo, cout), float("nan"), dtype=F32, device=DEV)
    ob = torch.empty((To, Ho, Wo, cout), dtype=BF, device=DEV)
    _ffi.call("wf_conv3d_cl", xd.data_ptr(), wk.data_ptr(), bd.data_ptr(), rd.data_ptr(), of.data_ptr(), ob.data_ptr(), T, H, W, cin,
              To, Ho, Wo, cout, k[0], k[1], k[2], st, ss, pt, ps, ps, 1 if up2 else 0, 0, zp.data_ptr(), ops.stream())
    want = ref + resid
    err = (of.cpu() - want).abs().max().item()
    assert err <= 2e-3 * max(1.0, want.abs().max().item()), err
    assert (ob.float().cpu() - want).abs().max().item() <= 1e-2 * want.abs().max().item()


@pytest.mark.parametrize("cfg", [
    dict(T=9, H=64, W=64, cin=96, cout=96),        # whole tiles
    dict(T=5, H=61, W=70, cin=96, cout=192),       # ragged rows and columns, two output-channel blocks
    dict(T=3, H=24, W=200, cin=192, cout=96),      # several column tiles, last one ragged
    dict(T=4, H=40, W=104, cin=384, cout=384),     # the 384-wide stage (24 channel slices, 4 output blocks)
    dict(T=6, H=16, W=64, cin=32, cout=32),        # thin layer: partial output-channel block
    dict(T=3, H=26, W=66, cin=96, cout=96, halo=True),   # row slab with halo rows (ph = 0, Hi = Ho + 2): the sharded VAE
])
def test_conv3d_333_vs_torch(cfg):
    """wf_conv3d_333 (LDS-resident patch kernel on re-packed weights) against the fp32 reference of the same op."""
    from worldforge_amd import _ffi, ops
    g = torch.Generator().manual_seed(5)
    T, H, W, cin, cout = cfg["T"], cfg["H"], cfg["W"], cfg["cin"], cfg["cout"]
    halo = cfg.get("halo", False)
    x = torch.randn(T, H, W, cin, generator=g).to(BF)
    w = (torch.randn(cout, cin, 3, 3, 3, generator=g) / math.sqrt(cin * 27)).to(BF)
    b = torch.randn(cout, generator=g) * 0.1
    ref = _conv_ref(x, w.float(), b, (3, 3, 3), 1, 1, 2, 1)          # [T, H, W, cout]
    if halo:
        ref = ref[:, 1:-1]                                           # the slab's own rows: input rows 0 and H-1 are halo
    Ho = ref.shape[1]
    resid = torch.randn(T, Ho, W, cout, generator=g)
    wk = w.permute(0, 2, 3, 4, 1).reshape(cout, 27, cin).contiguous().to(DEV)
    wp = torch.empty((27, cin // 16, cout, 16), dtype=BF, device=DEV)
    _ffi.call("wf_conv3d_pack333", wk.data_ptr(), wp.data_ptr(), cout, cin, ops.stream())
    assert torch.equal(wp.permute(2, 0, 1, 3).reshape(cout, 27, cin), wk)
    xd, bd, rd = x.to(DEV), b.to(DEV), resid.to(DEV)
    zp = torch.zeros(4096, dtype=BF, device=DEV)
    of = torch.full((T, Ho, W, cout), float("nan"), dtype=F32, device=DEV)
    ob = torch.empty((T, Ho, W, cout), dtype=BF, device=DEV)
    _ffi.call("wf_conv3d_333", xd.data_ptr(), wp.data_ptr(), bd.data_ptr(), rd.data_ptr(), of.data_ptr(), ob.data_ptr(), T, H, W, cin, Ho,
              cout, 0 if halo else 1, zp.data_ptr(), zp.numel() * 2, 0, cin, ops.stream())
    want = ref + resid
    err = (of.cpu() - want).abs().max().item()
    assert err <= 2e-3 * max(1.0, want.abs().max().item()), err
    assert (ob.float().cpu() - want).abs().max().item() <= 1e-2 * want.abs().max().item()
    # the epilogue every ResidualBlock conv of the VAE runs (bias, fp32 out only, with / without the residual) goes through LDS on full tiles
    # (round 4) where the call above stored straight from the accumulator layout: same bits
    for rd_ in (rd, None):
        of3 = torch.full_like(of, float("nan"))
        _ffi.call("wf_conv3d_333", xd.data_ptr(), wp.data_ptr(), bd.data_ptr(), rd_.data_ptr() if rd_ is not None else None, of3.data_ptr(), None,
                  T, H, W, cin, Ho, cout, 0 if halo else 1, zp.data_ptr(), zp.numel() * 2, 0, cin, ops.stream())
        if rd_ is not None:
            assert torch.equal(of3, of)
        else:             # without the residual: (acc + bias) alone -- compare with the same call through the direct-store form (bf16 copy requested)
            of4, ob4 = torch.full_like(of, float("nan")), torch.empty_like(ob)
            _ffi.call("wf_conv3d_333", xd.data_ptr(), wp.data_ptr(), bd.data_ptr(), None, of4.data_ptr(), ob4.data_ptr(), T, H, W, cin, Ho, cout,
                      0 if halo else 1, zp.data_ptr(), zp.numel() * 2, 0, cin, ops.stream())
            assert torch.equal(of3, of4)
    # same arithmetic as the generic kernel: fp32 accumulation order differs only within the MFMA k-steps
    of2 = torch.empty_like(of)
    _ffi.call("wf_conv3d_cl", xd.data_ptr(), wk.data_ptr(), bd.data_ptr(), rd.data_ptr(), of2.data_ptr(), None, T, H, W, cin, T, Ho, W, cout,
              3, 3, 3, 1, 1, 2, 0 if halo else 1, 1, 0, 0, zp.data_ptr(), ops.stream())
    assert (of - of2).abs().max().item() <= 1e-4 * max(1.0, want.abs().max().item())


def test_conv3d_tsplit_matches_upsample3d_interleave():
    from worldforge_amd import _ffi, ops
    g = torch.Generator().manual_seed(4)
    T, H, W, C = 3, 4, 5, 96
    x = torch.randn(T, H, W, C, generator=g).to(BF)
    w = (torch.randn(2 * C, C, 3, 1, 1, generator=g) / math.sqrt(3 * C)).to(BF)
    b = torch.randn(2 * C, generator=g) * 0.1
    y = _conv_ref(x, w.float(), b, (3, 1, 1), 1, 1, 2, 0)            # [T, H, W, 2C]
    want = torch.stack((y[..., :C], y[..., C:]), dim=1).reshape(2 * T, H, W, C)
    out = torch.zeros((1 + 2 * T, H, W, C), dtype=BF, device=DEV)
    wk = w.permute(0, 2, 3, 4, 1).reshape(2 * C, -1, C).contiguous().to(DEV)
    xd, bd = x.to(DEV), b.to(DEV)
    _ffi.call("wf_conv3d_cl", xd.data_ptr(), wk.data_ptr(), bd.data_ptr(), None, None, out.data_ptr(), T, H, W, C, T, H, W, 2 * C,
              3, 1, 1, 1, 1, 2, 0, 0, 0, 1, None, ops.stream())
    assert out[0].abs().max().item() == 0
    assert (out[1:].float().cpu() - want).abs().max().item() <= 1e-2 * want.abs().max().item()


def test_rms_silu_softmax_transpose_permute():
    from worldforge_amd import _ffi, ops
    g = torch.Generator().manual_seed(6)
    for C in (96, 192, 384):
        x = torch.randn(50, C, generator=g) * 2
        gam = 1 + 0.1 * torch.randn(C, generator=g)
        xd, gd = x.to(DEV), gam.to(DEV)
        ob = torch.empty((50, C), dtype=BF, device=DEV)
        of = torch.empty((50, C), dtype=F32, device=DEV)
        for silu in (0, 1):
            _ffi.call("wf_rms_silu_cl", xd.data_ptr(), gd.data_ptr(), ob.data_ptr(), of.data_ptr(), 50, C, silu, ops.stream())
            ref = F.normalize(x, dim=1) * C ** 0.5 * gam
            ref = F.silu(ref) if silu else ref
            assert (of.cpu() - ref).abs().max().item() <= 1e-5 * max(1, ref.abs().max().item())
            assert (ob.float().cpu() - ref).abs().max().item() <= 8e-3 * ref.abs().max().item()
    S = torch.randn(37, 200, generator=g) * 3
    Sd = S.to(DEV)
    P = torch.full((37, 208), float("nan"), dtype=BF, device=DEV)
    _ffi.call("wf_softmax_rows", Sd.data_ptr(), 200, P.data_ptr(), 208, 37, 200, 0.5, ops.stream())
    ref = torch.softmax(S * 0.5, dim=-1)
    assert (P[:, :200].float().cpu() - ref).abs().max().item() <= 4e-3 and P[:, 200:].abs().max().item() == 0
    A = torch.randn(70, 100, generator=g).to(BF)
    Ad = A.to(DEV)
    At = torch.full((40, 72), float("nan"), dtype=BF, device=DEV)
    _ffi.call("wf_transpose_bf16", Ad[:, 30:].data_ptr(), 100, At.data_ptr(), 72, 70, 40, ops.stream())
    assert torch.equal(At[:, :70].cpu(), A[:, 30:70].t()) and At[:, 70:].abs().max().item() == 0
    v = torch.randn(3, 2 * 4 * 5, generator=g)
    vd = v.to(DEV)
    cl = torch.empty((40, 3), dtype=F32, device=DEV)
    _ffi.call("wf_ncthw_to_cl", vd.data_ptr(), cl.data_ptr(), None, 3, 3, 40, ops.stream())
    assert torch.equal(cl.cpu(), v.t())
    back = torch.empty((3, 40), dtype=F32, device=DEV)
    _ffi.call("wf_cl_to_ncthw", cl.data_ptr(), back.data_ptr(), 3, 3, 40, 1.0, ops.stream())
    assert torch.equal(back.cpu(), v.clamp(-1, 1))
    clp = torch.full((40, 8), float("nan"), dtype=BF, device=DEV)
    _ffi.call("wf_ncthw_to_cl", vd.data_ptr(), None, clp.data_ptr(), 3, 8, 40, ops.stream())
    assert torch.equal(clp[:, :3].cpu(), v.t().to(BF)) and clp[:, 3:].abs().max().item() == 0
    back2 = torch.empty((2, 40), dtype=F32, device=DEV)
    _ffi.call("wf_cl_to_ncthw", cl.data_ptr(), back2.data_ptr(), 2, 3, 40, 0.0, ops.stream())
    assert torch.equal(back2.cpu(), v[:2])


def test_vae_protocol_and_errors(model):
    assert model.config.z_dim == 16 and len(model.config.latents_mean) == 16 and model.dtype == torch.float32
    assert model.temperal_downsample == [False, True, True]
    with pytest.raises(ValueError):
        model.encode(torch.zeros(1, 3, 4, 16, 16, device=DEV))


from tests.fakes import SimComm as _SimComm  # noqa: E402


# (2, 24): odd slab height at the decoder entry; (8, 96): 12 latent rows on 8 ranks = the low-resolution stage in 4 row groups of 2 ranks
@pytest.mark.parametrize("P,H", [(2, 64), (4, 64), (8, 64), (2, 24), (8, 96)])
def test_row_sharded_vae_equals_unsharded(P, H, model):
    import threading
    from worldforge_amd.vae import AutoencoderKLWan
    g = torch.Generator().manual_seed(21)
    Fr, Wd = 9, 48
    x = (torch.rand(1, 3, Fr, H, Wd, generator=g) * 2 - 1).to(DEV)
    z = torch.randn(1, 16, 3, H // 8, Wd // 8, generator=g).to(DEV)
    ref_mu = model.encode(x).latent_dist.mode()
    ref_dec = model.decode(z, return_dict=False)[0]
    shared = {"slots": [None] * P, "bar": threading.Barrier(P)}
    res, errs = [None] * P, []

    def worker(r):
        try:
            m = AutoencoderKLWan(DEV, comm=_SimComm(P, r, shared), precision=model.precision)
            m.w = model.w
            assert m.can_shard(H // 8)
            res[r] = (m.encode(x).latent_dist.mode(), m.decode(z, return_dict=False)[0])
        except Exception as e:  # pragma: no cover
            errs.append(e)
            shared["bar"].abort()

    th = [threading.Thread(target=worker, args=(r,)) for r in range(P)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    for r in range(P):
        assert torch.equal(res[r][0], ref_mu), (r, (res[r][0] - ref_mu).abs().max())
        assert torch.equal(res[r][1], ref_dec), (r, (res[r][1] - ref_dec).abs().max())


# ---- precision="fp32": three-term split operands (the reference runs the VAE in fp32, INFER:185-189) -----------------------------
@pytest.fixture(scope="module")
def model_fp32():
    from worldforge_amd.vae import AutoencoderKLWan
    m = AutoencoderKLWan(DEV, precision="fp32").load_state_dict(ovae.random_weights(seed=5))
    assert m.precision == "fp16x3" and m.f16 and m.x3      # "fp32" names the fp32-class default: fp16 three-term operands since round 4
    return m


@pytest.fixture(scope="module")
def model_bf16x3():
    from worldforge_amd.vae import AutoencoderKLWan
    return AutoencoderKLWan(DEV, precision="bf16x3").load_state_dict(ovae.random_weights(seed=5))


@pytest.mark.parametrize("side", [0, 1])
def test_split_f16x3_reconstructs_fp32(side):
    """fp16 parts: hi + lo reproduces x to 2^-22 relative while lo is a normal fp16 number, and to the fp16 subnormal quantum 2^-24
    absolutely below that; same layouts as the bf16 split."""
    from worldforge_amd import _ffi, ops
    g = torch.Generator().manual_seed(11)
    rows, C, ld = 37, 96, 128
    src = (torch.randn(rows, ld, generator=g) * torch.logspace(-3, 3, rows).unsqueeze(1)).to(DEV)
    dst = torch.empty(rows, 3 * C, dtype=torch.float16, device=DEV)
    _ffi.call("wf_split_f16x3", src.data_ptr(), ld, dst.data_ptr(), 3 * C, rows, C, side, ops.stream())
    x = src[:, :C]
    hi = x.to(torch.float16)
    lo = (x - hi.float()).to(torch.float16)
    want = torch.cat([hi, lo, hi] if side == 0 else [hi, hi, lo], dim=1)
    assert torch.equal(dst, want)
    err = ((hi.float() + lo.float()) - x).abs()
    assert bool((err <= torch.maximum(2.0 ** -22 * x.abs(), torch.tensor(2.0 ** -25, device=DEV))).all())


def test_f16_range_flag_is_raised_and_reported():
    """fp16 cannot hold |x| > 65504: the split raises the sticky device flag and the VAE turns it into an error (never a silent inf)."""
    import ctypes
    from worldforge_amd import _ffi, ops
    from worldforge_amd.vae import AutoencoderKLWan
    flag = ctypes.c_int(-1)
    _ffi.call("wf_f16_overflow_flag", ctypes.byref(flag), 1, ops.stream())
    src = torch.ones(4, 32, device=DEV)
    dst = torch.empty(4, 96, dtype=torch.float16, device=DEV)
    _ffi.call("wf_split_f16x3", src.data_ptr(), 32, dst.data_ptr(), 96, 4, 32, 0, ops.stream())
    _ffi.call("wf_f16_overflow_flag", ctypes.byref(flag), 1, ops.stream())
    assert flag.value == 0
    src[2, 5] = 7.0e4
    _ffi.call("wf_split_f16x3", src.data_ptr(), 32, dst.data_ptr(), 96, 4, 32, 0, ops.stream())
    _ffi.call("wf_f16_overflow_flag", ctypes.byref(flag), 0, ops.stream())
    assert flag.value == 1
    m = AutoencoderKLWan(DEV, precision="fp16x3")
    m._note_range("test")                           # what every encode / decode ends with: an asynchronous copy of the flag, no host sync
    with pytest.raises(RuntimeError, match="fp16 range"):
        m.check_range()                             # the explicit check point (schedulers / pipelines): reads and resets
    _ffi.call("wf_f16_overflow_flag", ctypes.byref(flag), 1, ops.stream())
    assert flag.value == 0
    m._note_range("clean")
    m.check_range()                                 # nothing raised since the reset
    # a flag raised by one call surfaces at the NEXT VAE call even without an explicit check (once its copy has landed)
    _ffi.call("wf_split_f16x3", src.data_ptr(), 32, dst.data_ptr(), 96, 4, 32, 0, ops.stream())
    m._note_range("first")
    torch.cuda.synchronize()
    with pytest.raises(RuntimeError, match="AutoencoderKLWan.first"):
        m._note_range("second")
    _ffi.call("wf_f16_overflow_flag", ctypes.byref(flag), 1, ops.stream())
    # weights: since round 5 every weight matrix is stored times a power of two that puts its largest magnitude into [2^13, 2^14) (undone in
    # the kernels' epilogues): a weight of 1e5 is no longer out of range -- only a non-finite one is refused
    w = ovae.random_weights(seed=5)
    k = next(k for k in w if k.endswith("residual.2.weight"))
    w[k] = w[k].clone()
    w[k].view(-1)[0] = 1.0e5
    mbig = AutoencoderKLWan(DEV, precision="fp16x3").load_state_dict(w)
    sc = mbig.w[k[:-len("weight")] + "w.scale"]
    assert sc == 2.0 ** 3 and abs(float(mbig.w[k[:-len("weight")] + "w"].float().abs().max()) - 1.0e5 / 8) <= 4.0   # 1e5 * 2^-3 = 12500 in [2^13, 2^14): an fp16 there is a multiple of 8
    w[k].view(-1)[0] = float("inf")
    with pytest.raises(ValueError, match="not finite"):
        AutoencoderKLWan(DEV, precision="fp16x3").load_state_dict(w)


def test_vae_decode_returns_while_the_gpu_is_still_busy(model_fp32):
    """No host synchronisation inside a VAE call for ITS OWN work (the fp16 range flag is copied asynchronously): with ~100 ms of GEMMs
    queued in front, decode() must come back to the host before the GPU has reached its work.  The NEXT VAE call is the deterministic
    check point of that flag (never timing-dependent) -- it waits for the previous call, not for its own."""
    from worldforge_amd import dit
    from worldforge_amd.vae import AutoencoderKLWan
    a = torch.randn(8192, 8192, device=DEV).to(torch.bfloat16)
    out = torch.empty(8192, 8192, dtype=torch.bfloat16, device=DEV)
    z = torch.randn(1, 16, 2, 8, 8, device=DEV)
    video = torch.rand(1, 3, 5, 64, 64, device=DEV) * 2 - 1
    model_fp32.decode(z, return_dict=False)          # warm (allocations, packed weights)
    model_fp32.encode(video)
    model_fp32.check_range()
    torch.cuda.synchronize()
    for _ in range(100):                             # ~1 ms each
        dit.gemm(a, a, None, out, dit.EPI_BF16)
    ev = torch.cuda.Event()
    model_fp32.decode(z, return_dict=False)
    ev.record()
    assert not ev.query(), "decode synchronised the host with the stream"
    assert model_fp32._flag_pending[1] == "decode"
    model_fp32.encode(video)                         # starts by waiting for decode's flag (read and cleared) ...
    assert model_fp32._flag_pending[1] == "encode"   # ... and leaves its own pending: nothing waited for encode itself
    model_fp32.check_range()                         # the explicit check point waits for the last call's flag
    assert model_fp32._flag_pending is None
    # strict_range=True: the check runs inside the call (one host synchronisation per call) -- for callers of the bare API
    strict = AutoencoderKLWan(DEV, precision="fp32", strict_range=True)
    strict.w = model_fp32.w
    for _ in range(20):
        dit.gemm(a, a, None, out, dit.EPI_BF16)
    strict.decode(z, return_dict=False)
    assert getattr(strict, "_flag_pending", None) is None      # read inside the call


def test_range_flag_of_a_vae_call_surfaces_at_the_next_call_deterministically(model_fp32):
    """a call that overflowed fp16 must fail at the NEXT VAE call whatever the timing (round 5 checked only if the flag's copy
    had already landed).  The overflow is injected through the producer the VAE itself uses."""
    import ctypes
    from worldforge_amd import _ffi, ops
    flag = ctypes.c_int(0)
    _ffi.call("wf_f16_overflow_flag", ctypes.byref(flag), 1, ops.stream())
    z = torch.randn(1, 16, 2, 8, 8, device=DEV)
    model_fp32.decode(z, return_dict=False)
    model_fp32.check_range()
    src = torch.full((4, 32), 7.0e4, device=DEV)
    dst = torch.empty(4, 96, dtype=torch.float16, device=DEV)
    a = torch.randn(8192, 8192, device=DEV).to(torch.bfloat16)
    out = torch.empty(8192, 8192, dtype=torch.bfloat16, device=DEV)
    from worldforge_amd import dit
    for _ in range(50):                              # the copy of the flag sits behind ~50 ms of work: it has NOT landed when the next call starts
        dit.gemm(a, a, None, out, dit.EPI_BF16)
    _ffi.call("wf_split_f16x3", src.data_ptr(), 32, dst.data_ptr(), 96, 4, 32, 0, ops.stream())
    model_fp32._note_range("decode")                 # what the overflowing call ends with
    with pytest.raises(RuntimeError, match="AutoencoderKLWan.decode"):
        model_fp32.encode(torch.rand(1, 3, 5, 64, 64, device=DEV) * 2 - 1)
    _ffi.call("wf_f16_overflow_flag", ctypes.byref(flag), 1, ops.stream())
    model_fp32.check_range()


@pytest.mark.parametrize("side", [0, 1])
def test_split_bf16x3_reconstructs_fp32(side):
    """hi + lo reproduces x to 2^-16 relative; layouts [hi | lo | hi] (activation) / [hi | hi | lo] (weight); strided source rows."""
    from worldforge_amd import _ffi, ops
    g = torch.Generator().manual_seed(11)
    rows, C, ld = 37, 96, 128
    src = (torch.randn(rows, ld, generator=g) * torch.logspace(-3, 3, rows).unsqueeze(1)).to(DEV)
    dst = torch.empty(rows, 3 * C, dtype=BF, device=DEV)
    _ffi.call("wf_split_bf16x3", src.data_ptr(), ld, dst.data_ptr(), 3 * C, rows, C, side, ops.stream())
    x = src[:, :C]
    hi = x.to(BF)
    lo = (x - hi.float()).to(BF)
    want = torch.cat([hi, lo, hi] if side == 0 else [hi, hi, lo], dim=1)
    assert torch.equal(dst, want)
    assert ((hi.float() + lo.float()) - x).abs().max().item() <= 2.0 ** -16 * x.abs().max().item()


# bars per fp32-class mode: (rel L2 of mu / dec, max abs on pixels in [-1, 1]) against the fp32 goldens
# (measured on MI355X, profiles/r6_tolerances.txt: fp16x3 2.6e-6 / 3.4e-6 / 8.8e-6 with the weight operands stored power-of-two scaled --
# round 4, unscaled: 4.2e-6 / 6.0e-6 / 1.6e-5 -- the level at which two fp32 implementations of the same network differ by accumulation
# order; bf16x3 1.9e-5 / 2.2e-5 / 5.2e-5; the bars sit at <= 2x)
FP32_CLASS_BARS = {"bf16x3": (4.4e-5, 1.04e-4), "fp16x3": (6.8e-6, 1.76e-5)}


@pytest.mark.parametrize("mode", ["fp16x3", "bf16x3"])
@pytest.mark.parametrize("name", CASES)
def test_fp32_mode_encode_decode_vs_twin_goldens(name, mode, model_fp32, model_bf16x3, golden_dir):
    """The fp32-class VAE against the fp32 twin goldens, both operand splits: fp16 parts (default: ~2^-22 per product) and bf16 parts
    (~2^-16).  What is left: the dropped lo.lo term, fp32 accumulation order and the transcendental error of SiLU / softmax."""
    from tests._tol import within
    model = model_fp32 if mode == "fp16x3" else model_bf16x3
    g = np.load(os.path.join(golden_dir, "g8_vae.npz"))
    x, z = torch.from_numpy(g[f"{name}_x"]), torch.from_numpy(g[f"{name}_z"])
    mu = model.encode(x.to(DEV)).latent_dist.mode().cpu()
    dec = model.decode(z.to(DEV), return_dict=False)[0].cpu()
    mu_ref, dec_ref = torch.from_numpy(g[f"{name}_mu"]), torch.from_numpy(g[f"{name}_dec"])
    e_mu, e_dec = _rel(mu, mu_ref), _rel(dec, dec_ref)
    print(f"[{mode} {name}] rel L2: mu {e_mu:.3e} dec {e_dec:.3e}; max abs: mu {(mu - mu_ref).abs().max():.3e} "
          f"dec {(dec - dec_ref).abs().max():.3e}")
    tol_rel, tol_abs = FP32_CLASS_BARS[mode]
    within(f"vae.{mode}.mu.rel_l2", e_mu, tol_rel)
    within(f"vae.{mode}.dec.rel_l2", e_dec, tol_rel)
    within(f"vae.{mode}.dec.max_abs", (dec - dec_ref).abs().max().item(), tol_abs)


@pytest.mark.parametrize("P,H", [(2, 64), (4, 64), (8, 96)])   # (8, 96): the low-resolution stage in 4 row groups of 2 ranks each
def test_fp32_mode_row_sharded_equals_unsharded(P, H, model_fp32):
    """The row-slab (multi-GPU) VAE in the fp32-class mode: P simulated ranks == one rank, bit for bit."""
    from tests.test_gpu_multirank import _run_ranks
    from worldforge_amd.vae import AutoencoderKLWan
    g = torch.Generator().manual_seed(21)
    x = (torch.rand(1, 3, 9, H, 96, generator=g) * 2 - 1).to(DEV)
    z = torch.randn(1, 16, 3, H // 8, 12, generator=g).to(DEV)
    mu0 = model_fp32.encode(x).latent_dist.mode().clone()
    dec0 = model_fp32.decode(z, return_dict=False)[0].clone()

    def run(comm):
        v = AutoencoderKLWan(DEV, comm=comm, precision="fp32")
        v.w = model_fp32.w
        assert v.can_shard(H // 8)
        return v.encode(x).latent_dist.mode().clone(), v.decode(z, return_dict=False)[0].clone()

    for r, (mu, dec) in enumerate(_run_ranks(P, run)):
        assert torch.equal(mu, mu0), (r, (mu - mu0).abs().max())
        assert torch.equal(dec, dec0), (r, (dec - dec0).abs().max())


@pytest.mark.parametrize("name", ["f9_32x32", "f5_48x40"])
def test_diffusers_layout_checkpoint_loads_and_matches_executed_class(name, golden_dir, tmp_path):
    """A diffusers-layout checkpoint on disk (vae/diffusion_pytorch_model.safetensors with the names + shapes of the class the reference
    executes, recorded in g8b) -> AutoencoderKLWan.from_pretrained -> encode / decode against that class's recorded outputs."""
    from safetensors.torch import save_file
    from worldforge_amd.vae import AutoencoderKLWan, diffusers_key_map
    b = np.load(os.path.join(golden_dir, "g8b_vae_akw.npz"))
    g = np.load(os.path.join(golden_dir, "g8_vae.npz"))
    shapes = {str(n): tuple(int(v) for v in str(s).split(",")) for n, s in zip(b["param_names"], b["param_shapes"])}
    W = ovae.random_weights(seed=5)
    inv = {v: k for k, v in diffusers_key_map().items()}
    sd = {}
    for k, v in W.items():
        base, _, leaf = k.rpartition(".")
        dk = f"{inv[base]}.{leaf}"
        sd[dk] = v.reshape(shapes[dk]).contiguous()
    assert set(sd) == set(shapes)
    os.makedirs(tmp_path / "vae")
    save_file(sd, str(tmp_path / "vae" / "diffusion_pytorch_model.safetensors"))
    for precision, tol_rel, tol_abs in (("bf16", 1.5e-2, 6e-2), ("fp32",) + FP32_CLASS_BARS["fp16x3"], ("bf16x3",) + FP32_CLASS_BARS["bf16x3"]):
        m = AutoencoderKLWan.from_pretrained(str(tmp_path), device=DEV, precision=precision)
        mu = m.encode(torch.from_numpy(g[f"{name}_x"]).to(DEV)).latent_dist.mode().cpu()
        dec = m.decode(torch.from_numpy(g[f"{name}_z"]).to(DEV), return_dict=False)[0].cpu()
        mu_ref, dec_ref = torch.from_numpy(b[f"{name}_mu"]), torch.from_numpy(b[f"{name}_dec"])
        assert _rel(mu, mu_ref) <= tol_rel and _rel(dec, dec_ref) <= tol_rel, (precision, _rel(mu, mu_ref), _rel(dec, dec_ref))
        assert (dec - dec_ref).abs().max().item() <= tol_abs


@pytest.mark.parametrize("cfg", [
    dict(T=2, H=10, W=12, cin=32, cout=8),          # small: the generic kernel (M < 64 x 512)
    dict(T=3, H=96, W=128, cin=64, cout=96),        # large enough for the ping-pong kernel
    dict(T=2, H=33, W=70, cin=96, cout=48),         # ragged tiles, partial output-channel block
])
def test_upsample_conv_as_four_phase_convs(cfg):
    """wf_conv3d_cl_scatter: the four 2 x 2 phase convolutions (taps pre-summed) == nearest-2x upsample + 3 x 3 conv (vae.py:76-86), and
    == the gathered form wf_conv3d_cl(up2 = 1) of the same bf16 operands up to the pre-summed weights' bf16 rounding."""
    from worldforge_amd import _ffi, ops
    g = torch.Generator().manual_seed(11)
    T, H, W, cin, cout = cfg["T"], cfg["H"], cfg["W"], cfg["cin"], cfg["cout"]
    x = torch.randn(T, H, W, cin, generator=g).to(BF)
    w = (torch.randn(cout, cin, 3, 3, generator=g) / math.sqrt(cin * 9))
    b = torch.randn(cout, generator=g) * 0.1
    xu = torch.nn.functional.interpolate(x.float().permute(0, 3, 1, 2), scale_factor=2, mode="nearest")
    ref = torch.nn.functional.conv2d(xu, w, b, padding=1).permute(0, 2, 3, 1)            # [T, 2H, 2W, cout]
    xd, bd = x.to(DEV), b.to(DEV)
    zp = torch.zeros(64, dtype=BF, device=DEV)
    out = torch.full((T, 2 * H, 2 * W, cout), float("nan"), dtype=F32, device=DEV)
    groups = (([0], [1, 2]), ([0, 1], [2]))
    keep = []
    for py in range(2):
        for px in range(2):
            wp = torch.stack([torch.stack([sum(w[:, :, dy, dx] for dy in groups[py][a] for dx in groups[px][bb]) for bb in range(2)], dim=-1)
                              for a in range(2)], dim=-2)
            wk = wp.permute(0, 2, 3, 1).reshape(cout, 4, cin).to(BF).contiguous().to(DEV)
            keep.append(wk)
            _ffi.call("wf_conv3d_cl_scatter", xd.data_ptr(), wk.data_ptr(), bd.data_ptr(), None, out.data_ptr(), None, T, H, W, cin, T, H, W, cout,
                      1, 2, 2, 1, 1, 0, 1 - py, 1 - px, zp.data_ptr(), 2 * H, 2 * W, 2, py, 2, px, ops.stream())
    got = out.cpu()
    assert torch.isfinite(got).all()                                                     # every output pixel written exactly by one phase
    tol = 1e-2 * max(1.0, ref.abs().max().item())                                        # bf16 weights (the sums are rounded once more)
    assert (got - ref).abs().max().item() <= tol
    w3 = w.permute(0, 2, 3, 1).reshape(cout, 9, cin).to(BF).contiguous().to(DEV)
    gathered = torch.empty_like(out)
    _ffi.call("wf_conv3d_cl", xd.data_ptr(), w3.data_ptr(), bd.data_ptr(), None, gathered.data_ptr(), None, T, H, W, cin, T, 2 * H, 2 * W, cout,
              1, 3, 3, 1, 1, 0, 1, 1, 1, 0, zp.data_ptr(), ops.stream())
    assert (gathered.cpu() - ref).abs().max().item() <= tol
    with pytest.raises(RuntimeError):   # a scattered grid that does not fit the output tensor is refused
        _ffi.call("wf_conv3d_cl_scatter", xd.data_ptr(), keep[0].data_ptr(), bd.data_ptr(), None, out.data_ptr(), None, T, H, W, cin, T, H, W, cout,
                  1, 2, 2, 1, 1, 0, 1, 1, zp.data_ptr(), 2 * H - 1, 2 * W, 2, 1, 2, 0, ops.stream())


# ---- precision="fp16": one fp16 term per operand = the multiplicand width of a TF32 convolution (round 6) -------------------------------
# The yardstick is the reference's own reduced-precision arithmetic: g8c = the executed class (autoencoder_kl_wan.py) run by eager
# PyTorch in bf16 on CPU, the dtype the LongCat entry loads it in (run_longcat_worldforge_single.py:205), recorded together with its
# distance from the class's fp32 run.
@pytest.fixture(scope="module")
def model_fp16():
    from worldforge_amd.vae import AutoencoderKLWan
    m = AutoencoderKLWan(DEV, precision="tf32").load_state_dict(ovae.random_weights(seed=5))
    assert m.precision == "fp16" and m.f16 and not m.x3 and m.wide and m.OP == torch.float16
    return m


def test_cast_f16_and_producers_raise_the_range_flag():
    import ctypes
    from worldforge_amd import _ffi, ops
    flag = ctypes.c_int(-1)
    _ffi.call("wf_f16_overflow_flag", ctypes.byref(flag), 1, ops.stream())
    g = torch.Generator().manual_seed(2)
    src = torch.randn(7, 64, generator=g).to(DEV)
    dst = torch.empty(7, 40, dtype=torch.float16, device=DEV)
    _ffi.call("wf_cast_f16", src.data_ptr(), 64, dst.data_ptr(), 40, 7, 40, ops.stream())      # strided source and destination rows
    assert torch.equal(dst, src[:, :40].to(torch.float16))
    _ffi.call("wf_f16_overflow_flag", ctypes.byref(flag), 1, ops.stream())
    assert flag.value == 0
    src[3, 9] = -7.0e4
    _ffi.call("wf_cast_f16", src.data_ptr(), 64, dst.data_ptr(), 40, 7, 40, ops.stream())
    _ffi.call("wf_f16_overflow_flag", ctypes.byref(flag), 1, ops.stream())
    assert flag.value == 1
    # the layout kernel at the VAE's input
    v = torch.randn(3, 50, generator=g).to(DEV)
    out = torch.empty(50, 32, dtype=torch.float16, device=DEV)
    _ffi.call("wf_ncthw_to_cl_f16", v.data_ptr(), None, out.data_ptr(), 3, 32, 50, ops.stream())
    assert torch.equal(out[:, :3], v.t().to(torch.float16)) and not out[:, 3:].any()
    v[1, 4] = float("nan")
    _ffi.call("wf_ncthw_to_cl_f16", v.data_ptr(), None, out.data_ptr(), 3, 32, 50, ops.stream())
    _ffi.call("wf_f16_overflow_flag", ctypes.byref(flag), 1, ops.stream())
    assert flag.value == 1
    # RMS-norm with the fp16 output, softmax with fp16 probabilities
    x = torch.randn(11, 384, generator=g).to(DEV)
    gam = (1 + 0.1 * torch.randn(384, generator=g)).to(DEV)
    o16 = torch.empty(11, 384, dtype=torch.float16, device=DEV)
    o32 = torch.empty(11, 384, device=DEV)
    _ffi.call("wf_rms_silu_cl_f16", x.data_ptr(), gam.data_ptr(), o16.data_ptr(), o32.data_ptr(), 11, 384, 0, ops.stream())
    assert torch.equal(o16, o32.to(torch.float16))
    S = torch.randn(5, 72, generator=g).to(DEV)
    P16 = torch.empty(5, 80, dtype=torch.float16, device=DEV)
    Pb = torch.empty(5, 80, dtype=BF, device=DEV)
    _ffi.call("wf_softmax_rows_f16", S.data_ptr(), 72, P16.data_ptr(), 80, 5, 70, 0.5, ops.stream())
    _ffi.call("wf_softmax_rows", S.data_ptr(), 72, Pb.data_ptr(), 80, 5, 70, 0.5, ops.stream())
    want = torch.softmax(S[:, :70] * 0.5, dim=-1)
    assert (P16[:, :70].float() - want).abs().max().item() <= 2.0 ** -11 and not P16[:, 70:].any()
    assert (P16[:, :70].float() - want).abs().max().item() < (Pb[:, :70].float() - want).abs().max().item()


@pytest.mark.parametrize("name", CASES)
def test_fp16_mode_is_closer_to_fp32_than_the_references_bf16_run(name, model_fp16, model, golden_dir):
    """One-term fp16 ("fp16" / "tf32") and one-term bf16 ("bf16") operands with fp32 accumulation and an f32 stream, against the fp32
    goldens of the executed class (g8b) -- and against what the reference's OWN bf16 VAE (the LongCat entry's dtype) leaves, g8c."""
    from tests._tol import within
    g = np.load(os.path.join(golden_dir, "g8_vae.npz"))
    g32 = np.load(os.path.join(golden_dir, "g8b_vae_akw.npz"))
    gbf = np.load(os.path.join(golden_dir, "g8c_vae_akw_bf16.npz"))
    x, z = torch.from_numpy(g[f"{name}_x"]).to(DEV), torch.from_numpy(g[f"{name}_z"]).to(DEV)
    mu_ref, dec_ref = torch.from_numpy(g32[f"{name}_mu"]), torch.from_numpy(g32[f"{name}_dec"])
    ref_mu_rel, ref_dec_rel = float(gbf[f"{name}_mu_rel"]), float(gbf[f"{name}_dec_rel"])
    # the fixture's own figure is what its arrays say
    assert abs(_rel(torch.from_numpy(gbf[f"{name}_mu"]), mu_ref) - ref_mu_rel) <= 1e-6 * ref_mu_rel + 1e-9
    assert 1e-2 <= ref_mu_rel <= 2.5e-2 and 1e-2 <= ref_dec_rel <= 3e-2
    out = {}
    for mode, m in (("fp16", model_fp16), ("bf16", model)):
        mu = m.encode(x).latent_dist.mode().cpu()
        dec = m.decode(z, return_dict=False)[0].cpu()
        m.check_range()
        out[mode] = (_rel(mu, mu_ref), _rel(dec, dec_ref), (dec - dec_ref).abs().max().item())
        print(f"[{mode} {name}] rel L2 from fp32: mu {out[mode][0]:.3e} dec {out[mode][1]:.3e} max abs dec {out[mode][2]:.3e}   "
              f"(the reference's bf16 run: mu {ref_mu_rel:.3e} dec {ref_dec_rel:.3e})")
        # no farther from fp32 than the reference's own bf16 arithmetic is
        assert out[mode][0] <= ref_mu_rel and out[mode][1] <= ref_dec_rel, (mode, out[mode])
    # measured on MI355X (profiles/r6_tolerances.txt); bars at <= 2x
    within(f"vae.fp16.mu.rel_l2.{name}", out["fp16"][0], FP16_BARS[name][0])
    within(f"vae.fp16.dec.rel_l2.{name}", out["fp16"][1], FP16_BARS[name][1])
    within(f"vae.fp16.dec.max_abs.{name}", out["fp16"][2], FP16_BARS[name][2])
    # 11 significand bits against 8: the one-term fp16 mode is several times closer to fp32 than the one-term bf16 mode
    assert out["fp16"][0] < 0.25 * out["bf16"][0] and out["fp16"][1] < 0.25 * out["bf16"][1]


# per-case bars of the one-term fp16 mode: (mu rel L2, dec rel L2, dec max abs) at 2x what an MI355X measures (profiles/r6_a_pytest.log:
# 1.03e-3 / 1.39e-3 / 3.19e-3, 1.03e-3 / 1.33e-3 / 2.87e-3, 1.04e-3 / 1.18e-3 / 1.36e-3, 1.21e-3 / 1.43e-3 / 2.92e-3) -- 15x closer to the
# fp32 network than the reference's own bf16 module (1.5e-2 / 2.1e-2, g8c) and 7.5x closer than one-term bf16 operands
FP16_BARS = {"f9_32x32": (2.06e-3, 2.78e-3, 6.4e-3), "f5_48x40": (2.07e-3, 2.65e-3, 5.75e-3), "f1_32x32": (2.08e-3, 2.35e-3, 2.72e-3),
             "f17_16x24": (2.41e-3, 2.86e-3, 5.84e-3)}


@pytest.mark.parametrize("P,H", [(2, 64), (8, 96)])
def test_fp16_mode_row_sharded_equals_unsharded(P, H, model_fp16):
    from tests.test_gpu_multirank import _run_ranks
    from worldforge_amd.vae import AutoencoderKLWan
    g = torch.Generator().manual_seed(21)
    x = (torch.rand(1, 3, 9, H, 96, generator=g) * 2 - 1).to(DEV)
    z = torch.randn(1, 16, 3, H // 8, 12, generator=g).to(DEV)
    mu0 = model_fp16.encode(x).latent_dist.mode().clone()
    dec0 = model_fp16.decode(z, return_dict=False)[0].clone()

    def run(comm):
        v = AutoencoderKLWan(DEV, comm=comm, precision="fp16")
        v.w = model_fp16.w
        assert v.can_shard(H // 8)
        return v.encode(x).latent_dist.mode().clone(), v.decode(z, return_dict=False)[0].clone()

    for r, (mu, dec) in enumerate(_run_ranks(P, run)):
        assert torch.equal(mu, mu0), (r, (mu - mu0).abs().max())
        assert torch.equal(dec, dec0), (r, (dec - dec0).abs().max())


# ---- decoding only the pixel columns an IRR injection's blend can see (round 6) ----------------------------------------------------------
def _hole_mask(Fr, H, W, kind):
    """Masks in [0, 1] whose value is EXACTLY 1 outside a hole: 'right' = SURVEY 8d's growing hole with a sine-softened edge."""
    xs = torch.arange(W).view(1, 1, 1, 1, W).float()
    fr = torch.arange(Fr).view(1, 1, Fr, 1, 1).float() / max(Fr - 1, 1)
    if kind == "right":
        edge = W * (1 - 0.35 * fr)
        d = (edge - xs).clamp(min=0)
        return (torch.sin(math.pi / 2 * (d / 15).clamp(0, 1)) * (xs < edge)).expand(1, 1, Fr, H, W).contiguous()
    if kind == "middle":
        m = torch.ones(1, 1, Fr, H, W)
        m[:, :, 1:, H // 4:H // 2, W // 2 - 20:W // 2 + 9] = 0.25
        return m
    if kind == "none":
        return torch.ones(1, 1, Fr, H, W)
    return torch.rand(1, 1, Fr, H, W, generator=torch.Generator().manual_seed(3))          # holes everywhere


def test_mask_column_range_kernel():
    import ctypes  # noqa: F401
    from worldforge_amd import _ffi, ops
    for kind, W in (("right", 256), ("middle", 200), ("none", 64), ("rand", 72)):
        m = _hole_mask(5, 24, W, kind).to(DEV)
        out = torch.empty(2, dtype=torch.int32, device=DEV)
        _ffi.call("wf_mask_column_range", m.data_ptr(), m.numel() // W, W, out.data_ptr(), ops.stream())
        cols = (m != 1).flatten(0, 3).any(0).nonzero().flatten()
        want = (int(cols.min()), int(cols.max()) + 1) if cols.numel() else (W, 0)
        assert tuple(int(v) for v in out.cpu()) == want, (kind, out, want)
    m = torch.ones(2, 40, device=DEV)
    m[1, 17] = float("nan")                      # a NaN is "not 1": the column must be decoded
    out = torch.empty(2, dtype=torch.int32, device=DEV)
    _ffi.call("wf_mask_column_range", m.data_ptr(), 2, 40, out.data_ptr(), ops.stream())
    assert tuple(int(v) for v in out.cpu()) == (17, 18)


@pytest.mark.parametrize("kind", ["right", "middle", "none", "rand"])
@pytest.mark.parametrize("prec", ["fp16x3", "bf16"])
def test_decode_of_the_needed_columns_gives_the_same_blend_bit_for_bit(kind, prec, model, model_fp32):
    """The decoded video's only consumer in fuse_latents is the blend (SCHED:1380): decoding just the columns with a mask value != 1 (+ the
    receptive-field halo, whole frames through the latent-resolution stage) must give the SAME fused video and the same posterior as
    decoding everything -- bit for bit -- and the needed columns of the decoded video itself must be identical."""
    from worldforge_amd import ops
    m0 = model_fp32 if prec == "fp16x3" else model
    g = torch.Generator().manual_seed(31)
    Fr, H, W = 9, 64, 384                                        # 48 latent columns
    z = torch.randn(1, 16, 3, H // 8, W // 8, generator=g).to(DEV)
    ref = torch.rand(1, 3, Fr, H, W, generator=g).to(DEV)
    mask = _hole_mask(Fr, H, W, kind).to(DEV)
    full = m0.decode(z, return_dict=False)[0]
    cols = m0.needed_columns(mask)
    crop = m0._crop_range(cols, W // 8)
    part = m0.decode(z, return_dict=False, columns=cols)[0]
    if kind == "right":
        assert cols == (29, 48) and crop == (16, 48)               # mask != 1 from pixel 235 (0.65 * 384 - 15 = 234.6) -> latent column 29
    elif kind == "middle":
        assert cols == (21, 26) and crop == (8, 40)
    elif kind == "none":
        assert cols == (0, 0) and crop == (0, 16)                  # nothing can reach the result: the cheapest crop
    else:
        assert crop is None and torch.equal(part, full)            # holes everywhere: everything is decoded
    if crop is not None:
        c0, c1 = cols
        assert torch.equal(part[..., 8 * c0:8 * c1], full[..., 8 * c0:8 * c1])
        assert not torch.equal(part, full) or kind == "none"
        assert torch.isfinite(part).all()
    fused_full, fused_part = ops.blend_pixels(ref, mask, full), ops.blend_pixels(ref, mask, part)
    assert torch.equal(fused_full, fused_part)
    # the whole round trip through the entry point the schedulers use
    m0.crop_to_mask = False
    mu_full = m0.decode_blend_encode(z, ref, mask).mode().clone()
    m0.crop_to_mask = True
    try:
        mu_part = m0.decode_blend_encode(z, ref, mask).mode().clone()
    finally:
        m0.crop_to_mask = True
    m0.check_range()
    assert torch.equal(mu_full, mu_part)


@pytest.mark.parametrize("P,H", [(2, 64), (8, 96)])
def test_row_sharded_round_trip_with_cropped_decode_equals_unsharded(P, H, model_fp32):
    from tests.test_gpu_multirank import _run_ranks
    from worldforge_amd.vae import AutoencoderKLWan
    g = torch.Generator().manual_seed(33)
    Fr, W = 9, 256
    z = torch.randn(1, 16, 3, H // 8, W // 8, generator=g).to(DEV)
    ref = torch.rand(1, 3, Fr, H, W, generator=g).to(DEV)
    mask = _hole_mask(Fr, H, W, "right").to(DEV)
    model_fp32.crop_to_mask = False
    mu0 = model_fp32.decode_blend_encode(z, ref, mask).mode().clone()
    model_fp32.crop_to_mask = True
    assert model_fp32._crop_range(model_fp32.needed_columns(mask), W // 8) is not None

    def run(comm):
        v = AutoencoderKLWan(DEV, comm=comm, precision="fp32")
        v.w = model_fp32.w
        assert v.can_shard(H // 8) and v.crop_to_mask
        return v.decode_blend_encode(z, ref, mask).mode().clone()

    for r, mu in enumerate(_run_ranks(P, run)):
        assert torch.equal(mu, mu0), (r, (mu - mu0).abs().max())
