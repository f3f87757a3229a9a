"""GPU: every scheduler / injection kernel of the C-ABI against the CPU oracle (torch eager semantics), bit-exact
where the arithmetic is element-wise, tight tolerance where a reduction order differs."""
import numpy as np
import pytest
import torch

from oracle import inject as oinj
from oracle import sched as osch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _r(shape, seed, dtype=torch.float32, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(shape, generator=g) * scale).to(dtype)


def _eq(got, want):
    assert got.dtype == want.dtype, (got.dtype, want.dtype)
    np.testing.assert_array_equal(got.float().cpu().numpy(), want.float().numpy())


SHAPES = [(1, 16, 3, 4, 4), (1, 16, 21, 60, 104), (2, 16, 1, 3, 5), (1, 16, 0, 4, 4)]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_cfg_combine(shape, dt):
    from worldforge_amd import ops
    a, b = _r(shape, 1, dt), _r(shape, 2, dt)
    _eq(ops.cfg_combine(a.to(DEV), b.to(DEV), 4.0), oinj.cfg_combine(a, b, 4.0))


@pytest.mark.parametrize("ds", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("dv", [torch.float32, torch.bfloat16])
def test_x0_from_v(ds, dv):
    from worldforge_amd import ops
    s, v = _r(SHAPES[1], 3, ds), _r(SHAPES[1], 4, dv)
    sigma = torch.tensor(0.8996, dtype=torch.float32)
    from worldforge_amd.scheduler import scalar_as
    _eq(ops.x0_from_v(s.to(DEV), v.to(DEV), scalar_as(sigma, dv)), s - sigma * v)


@pytest.mark.parametrize("dx", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("dm0", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("dm1", [torch.float32, torch.bfloat16, None])
def test_unipc_update(dx, dm0, dm1):
    from worldforge_amd import ops
    st = osch.make_state(10, 3.0)
    st.step_index = 4
    x, m0 = _r(SHAPES[0], 5, dx), _r(SHAPES[0], 6, dm0)
    order = 1 if dm1 is None else 2
    m1 = None if dm1 is None else _r(SHAPES[0], 7, dm1)
    st.model_outputs = [m1, m0]
    want = osch.unip_update(st, x, order)
    from worldforge_amd.scheduler import scalar_as
    c1, c2, c3, rk = osch.unip_coeffs(st, order)
    res = torch.bfloat16 if (m1 is not None and dx == dm0 == dm1 == torch.bfloat16) else torch.float32
    got = ops.unipc_update(x.to(DEV), m0.to(DEV), None if m1 is None else m1.to(DEV), scalar_as(c1, dx), scalar_as(c2, dm0),
                           scalar_as(c3, res), 1.0 if rk is None else rk.item())
    _eq(got, want)


def test_unipc_final_step_sigma_zero():
    """sigma_t = 0: lambda = +inf, expm1(-inf) = -1 (SCHED:1019-1061) -> x_t = m0."""
    from worldforge_amd import ops
    st = osch.make_state(4, 3.0)
    st.step_index = 3
    x, m0 = _r(SHAPES[0], 8), _r(SHAPES[0], 9)
    st.model_outputs = [None, m0]
    want = osch.unip_update(st, x, 1)
    c1, c2, c3, _ = osch.unip_coeffs(st, 1)
    _eq(ops.unipc_update(x.to(DEV), m0.to(DEV), None, c1.item(), c2.item(), c3.item(), 1.0), want)
    assert torch.isfinite(want).all()


@pytest.mark.parametrize("dx", [torch.float32, torch.bfloat16])
def test_add_noise(dx):
    from worldforge_amd import ops
    st = osch.make_state(10, 3.0)
    x0, nz = _r(SHAPES[1], 10, dx), _r(SHAPES[1], 11)
    want = osch.add_noise_resample(st, x0, nz, st.resample_timesteps[3])
    s = st.resample_sigmas.to(dx)[3]
    _eq(ops.add_noise(x0.to(DEV), nz.to(DEV), (1 - s).item(), s.item()), want)


@pytest.mark.parametrize("dz", [torch.float32, torch.bfloat16])
def test_latent_affine(dz):
    from tests.fakes import VAE_MEAN, VAE_STD
    from worldforge_amd import ops
    z = _r(SHAPES[1], 12, dz, 2.0)
    _eq(ops.latent_denorm(z.to(DEV), VAE_MEAN, VAE_STD), oinj.latent_denorm(z, VAE_MEAN, VAE_STD))
    mu = _r(SHAPES[1], 13, torch.float32, 2.0)
    _eq(ops.latent_norm(mu.to(DEV), VAE_MEAN, VAE_STD, dz), oinj.latent_norm(mu, VAE_MEAN, VAE_STD, dz))


@pytest.mark.parametrize("shape", [(1, 3, 9, 32, 48), (1, 3, 5, 17, 23), (2, 3, 2, 8, 8)])
def test_blend_pixels(shape):
    from worldforge_amd import ops
    g = torch.Generator().manual_seed(14)
    ref = torch.rand(shape, generator=g)
    dec = torch.rand(shape, generator=g) * 2 - 1
    mask = torch.rand((shape[0], 1) + shape[2:], generator=g)
    mask[mask < 0.3] = 0
    mask[mask > 0.7] = 1
    _eq(ops.blend_pixels(ref.to(DEV), mask.to(DEV), dec.to(DEV)), oinj.blend_pixels(ref, mask, dec))


def test_postprocess_and_cast():
    from worldforge_amd import ops
    x = _r((3, 5, 16, 24), 15) * 0.8
    want = (x / 2 + 0.5).clamp(0, 1).permute(1, 2, 3, 0).contiguous()
    _eq(ops.postprocess_video(x.to(DEV)), want)
    y = _r(SHAPES[1], 16)
    _eq(ops.cast(y.to(DEV), torch.bfloat16), y.to(torch.bfloat16))


def test_channel_swap():
    from worldforge_amd import ops
    enc, pred = _r(SHAPES[0], 17), _r(SHAPES[0], 18, torch.bfloat16)
    want = enc.clone()
    for c in (2, 7, 15):
        want[:, c] = pred[:, c]
    got = ops.channel_swap_(enc.to(DEV).contiguous(), pred.to(DEV), [2, 7, 15])
    _eq(got, want)
    _eq(ops.channel_swap_(enc.to(DEV).contiguous(), pred.to(DEV), []), enc)


@pytest.mark.parametrize("sizes", [((20, 30), (32, 48)), ((64, 80), (32, 40)), ((7, 9), (7, 9)), ((5, 5), (13, 4))])
def test_resize(sizes):
    import torch.nn.functional as F
    from worldforge_amd import ops
    (hi, wi), (ho, wo) = sizes
    x = torch.rand(2, 3, 4, hi, wi, generator=torch.Generator().manual_seed(19))
    want = F.interpolate(x.reshape(-1, 1, hi, wi), size=(ho, wo), mode="bilinear", align_corners=False).reshape(2, 3, 4, ho, wo)
    got = ops.resize_bilinear2d(x.to(DEV), ho, wo).cpu()
    np.testing.assert_allclose(got.numpy(), want.numpy(), atol=2e-6, rtol=0)
    wantn = F.interpolate(x.reshape(-1, 1, hi, wi), size=(ho, wo), mode="nearest").reshape(2, 3, 4, ho, wo)
    _eq(ops.resize_nearest2d(x.to(DEV), ho, wo), wantn)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [SHAPES[0], SHAPES[1]])
def test_dsg(dt, shape):
    from worldforge_amd import ops
    w = _r(shape, 20, dt)
    g = (w.float() * 0.9 + 0.3 * _r(shape, 21)).to(dt)
    want = oinj.dsg(g, w, 4.0)
    got = ops.dsg(g.to(DEV), w.to(DEV), 4.0).cpu()
    assert got.dtype == want.dtype
    if dt == torch.float32:
        np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=2e-5, atol=2e-6)
    else:
        # scalar coefficients may land on the neighbouring bf16 value (different summation order): <= 1 bf16 ulp effect
        err = (got.float() - want.float()).abs().max().item()
        assert err <= 2.0 ** -7 * want.float().abs().max().item(), err


def test_dsg_identical_inputs_is_identity_direction():
    from worldforge_amd import ops
    g = _r(SHAPES[0], 22)
    got = ops.dsg(g.to(DEV), g.to(DEV), 4.0).cpu()
    want = oinj.dsg(g, g, 4.0)
    np.testing.assert_allclose(got.numpy(), want.numpy(), rtol=1e-4, atol=1e-4)


def test_flow_metrics_golden(golden_dir):
    import os
    from worldforge_amd import ops
    g = np.load(os.path.join(golden_dir, "g4_flf.npz"))
    for k in range(5):
        r, c = torch.from_numpy(g[f"fm{k}_ref"]), torch.from_numpy(g[f"fm{k}_chan"])
        sim = ops.flow_metrics(r.to(DEV), c.to(DEV)).cpu().item()
        assert sim == pytest.approx(float(g[f"fm{k}_sim"][0]), abs=2e-6), k


def test_flf_selector_end_to_end_golden(golden_dir):
    import os
    from worldforge_amd.flf import VideoMotionPCASelector
    g = np.load(os.path.join(golden_dir, "g4_flf.npz"))
    pred, enc = torch.from_numpy(g["e2e_pred"]).to(DEV), torch.from_numpy(g["e2e_enc"]).to(DEV)
    sel = VideoMotionPCASelector(flow_backend="tdiff")  # g4 was recorded without cv2: the temporal-difference branch
    np.testing.assert_allclose(sel.channel_similarities(pred, enc), g["e2e_sims"], atol=2e-6)
    for step in (0, 1, 3, 8, 12):
        want = g[f"e2e_step{step}"].tolist() if f"e2e_step{step}" in g else []
        assert sel.select_motion_related_channels(pred, enc, current_step=step) == want
    for si in range(5):
        for step in (0, 1, 2, 3, 5, 6, 10, 11, 30):
            assert sel.select_from_similarities(g[f"sel{si}_sims"], step) == g[f"sel{si}_step{step}"].tolist()


def test_errors_are_loud():
    from worldforge_amd import _ffi, ops
    with pytest.raises(RuntimeError):
        ops.cfg_combine(torch.zeros(4), torch.zeros(4), 1.0)  # CPU tensors: no fallback
    rc = _ffi.lib().wf_latent_affine(None, 0, None, 0, None, None, 0, 1, 16, 4, None)
    assert rc != 0 and b"null" in _ffi.lib().wf_last_error()


def test_soften_mask_gpu_matches_reference_goldens(golden_dir):
    """wf_soften_mask against the goldens recorded from the reference's soften_mask (infer_worldforge.py:105-150): disc and
    half-plane masks x 4 decay types x 2 transition distances.  The ramp is evaluated in double on both sides; the device sin / cos /
    exp may differ from libm in the last double bit, i.e. by at most one float32 ulp after the final rounding."""
    import os
    import numpy as np
    from worldforge_amd import ops
    g = np.load(os.path.join(golden_dir, "g10_harness.npz"))
    masks = torch.from_numpy(g["masks"].astype(np.float32)).to(DEV)
    for decay in ("linear", "exponential", "sine", "cosine"):
        for d in (5, 15):
            want = torch.from_numpy(g[f"soft_{decay}_{d}"])
            got = ops.soften_mask(masks, d, decay).cpu()
            diff = (got - want).abs().max().item()
            assert diff <= 1.2e-7, (decay, d, diff)
            if decay == "linear":
                assert torch.equal(got, want)
    with pytest.raises(ValueError):
        ops.soften_mask(masks, 5, "bogus")
    # soft (non-binary) input values, all-ones and all-zeros frames pass through untouched
    m = torch.ones(3, 40, 50)
    m[1] = 0
    m[2] = 0.25
    m[2, 10:20, 10:20] = 0
    out = ops.soften_mask(m.to(DEV), 15, "sine").cpu()
    assert torch.equal(out[0], m[0]) and torch.equal(out[1], m[1])
    from worldforge_amd import harness
    assert np.abs(out.numpy() - harness.soften_mask(m.numpy(), 15, "sine")).max() <= 1.2e-7
