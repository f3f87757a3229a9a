"""CPU: the Farneback restatement (oracle/farneback.py, parity UNPINNED -- no cv2 in the image) checked against the
algorithm's defining properties: it recovers known synthetic translations, gives zero flow for identical frames, follows
OpenCV's pyramid sizing rule, and its polynomial expansion reproduces the coefficients of an exact quadratic."""
import numpy as np
import pytest

from oracle import farneback as fb


def _texture(h, w, seed, smooth=4.0):
    rng = np.random.default_rng(seed)
    big = rng.standard_normal((h + 32, w + 32))
    k = np.exp(-0.5 * (np.arange(-12, 13) / smooth) ** 2)
    k /= k.sum()
    big = np.apply_along_axis(lambda r: np.convolve(r, k, mode="same"), 1, big)
    big = np.apply_along_axis(lambda r: np.convolve(r, k, mode="same"), 0, big)
    big = (big - big.min()) / (big.max() - big.min())
    return big


def _crop(big, h, w, oy, ox):
    return (big[16 + oy:16 + oy + h, 16 + ox:16 + ox + w] * 255).astype(np.uint8)


def test_pyramid_plan_follows_the_32_pixel_rule():
    assert [p[3:] for p in fb.pyramid_plan(60, 104)] == [(104, 60)]                       # C2 latents: one level
    assert [p[3:] for p in fb.pyramid_plan(58, 104)] == [(104, 58)]                       # C1
    assert [p[3:] for p in fb.pyramid_plan(90, 160)] == [(80, 45), (160, 90)]             # C3: two levels
    assert [p[3:] for p in fb.pyramid_plan(480, 832)] == [(104, 60), (208, 120), (416, 240), (832, 480)]
    plan = fb.pyramid_plan(90, 160)
    assert plan[0][1] == 0.5 and plan[0][2] == 3 and plan[1][1] == 0.0 and plan[1][2] == 3


def test_gaussian_kernels():
    assert fb.gaussian_kernel(3, 0.0).tolist() == [0.25, 0.5, 0.25]
    k = fb.gaussian_kernel(3, 0.5).astype(np.float64)
    e = np.exp(-2.0)
    assert np.allclose(k, np.array([e, 1, e]) / (1 + 2 * e), atol=1e-7)


def test_poly_exp_recovers_an_exact_quadratic():
    h, w = 40, 48
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    yc, xc = y - 20, x - 24
    img = (3.0 + 0.5 * xc - 0.25 * yc + 0.02 * xc * xc + 0.01 * yc * yc - 0.015 * xc * yc).astype(np.float32)
    R = fb.poly_exp(img)
    c = R[20, 24]                       # order: (y, x, y^2, x^2, xy) coefficients at the expansion point
    assert np.allclose(c, [-0.25, 0.5, 0.01, 0.02, -0.015], atol=2e-4)


@pytest.mark.parametrize("h,w,dy,dx", [(60, 104, 0, 1), (60, 104, 1, -1), (90, 160, -1, 2), (64, 64, 0, 0)])
def test_recovers_a_synthetic_translation(h, w, dy, dx):
    big = _texture(h, w, seed=h + 7 * dx + dy)
    a = _crop(big, h, w, 0, 0)
    b = _crop(big, h, w, -dy, -dx)      # content moves by (+dx, +dy)
    flow = fb.calc_optical_flow_farneback(a, b)
    assert flow.shape == (h, w, 2) and flow.dtype == np.float32
    inner = flow[12:-12, 12:-12]
    med = np.median(inner.reshape(-1, 2), axis=0)
    assert abs(med[0] - dx) <= 0.25 and abs(med[1] - dy) <= 0.25, med


def test_channel_flow_layout_and_quantisation():
    rng = np.random.default_rng(3)
    ch = rng.standard_normal((4, 20, 24)).astype(np.float32)
    gmin, grange = np.float32(ch.min()), np.float32(ch.max() - ch.min() + np.float32(1e-8))
    q = fb.quantise_channel(ch, gmin, grange)
    assert q.dtype == np.uint8 and q.min() == 0 and q.max() in (254, 255)
    fl = fb.channel_flow(ch, gmin, grange)
    assert fl.shape == (3, 2, 20, 24) and np.isfinite(fl).all()
