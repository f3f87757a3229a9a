"""CPU restatement of the dense optical flow the FLF gate uses when OpenCV is installed -- TEST INFRASTRUCTURE ONLY.

**Parity unpinned.**  The reference calls `cv2.calcOpticalFlowFarneback(gray1, gray2, None, pyr_scale=0.5, levels=3, winsize=15,
iterations=3, poly_n=5, poly_sigma=1.2, flags=0)` (wan_for_worldforge/utils/scheduling_unipc_multistep_clean.py:220-224) from
`opencv-python` (requirements.txt:8, version not pinned).  OpenCV is neither under /root/reference nor installed in this image and
the reference holds no test or golden vector for this call, so this module restates OpenCV's published algorithm
(G. Farneback, "Two-frame motion estimation based on polynomial expansion", SCIA 2003; OpenCV 4.x
modules/video/src/optflowgf.cpp: FarnebackPrepareGaussian / FarnebackPolyExp / FarnebackUpdateMatrices /
FarnebackUpdateFlow_Blur and FarnebackOpticalFlowImpl::calc, plus the GaussianBlur / resize conventions it relies on) from
the algorithm's definition, with the same float / double split as the C++ (float images and matrices, double window sums and
normal-equation solve).  It has NOT been compared with a real cv2; the known-answer tests (tests/test_oracle_farneback.py)
only establish that it recovers synthetic translations.  Everything else in the FLF gate (quantisation to uint8, per-channel
loop, metric, thresholds) is pinned through the reference import (oracle/inject.py).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this package.
"""
from __future__ import annotations


import numpy as np

F32 = np.float32
POLY_N, POLY_SIGMA, WINSIZE, ITERS, LEVELS, PYR_SCALE = 5, 1.2, 15, 3, 3, 0.5
MIN_SIZE = 32                      # optflowgf.cpp: pyramid stops when a level would be narrower / lower than 32 pixels
BORDER = (0.14, 0.14, 0.4472, 0.4472, 0.4472)


def cv_round(v: float) -> int:
    """cvRound: round half to even (lrint)."""
    return int(np.rint(v))


# ---------------------------------------------------------------------------------------------------------------------
# OpenCV conventions the flow relies on
# ---------------------------------------------------------------------------------------------------------------------
def gaussian_kernel(ksize: int, sigma: float) -> np.ndarray:
    """cv::getGaussianKernel(ksize, sigma, CV_32F): fixed table for sigma <= 0 and small odd sizes, else sampled + normalised."""
    small = {1: [1.0], 3: [0.25, 0.5, 0.25], 5: [0.0625, 0.25, 0.375, 0.25, 0.0625],
             7: [0.03125, 0.109375, 0.21875, 0.28125, 0.21875, 0.109375, 0.03125]}
    if sigma <= 0 and ksize in small:
        k = np.array(small[ksize], dtype=np.float64)
    else:
        s = sigma if sigma > 0 else ((ksize - 1) * 0.5 - 1) * 0.3 + 0.8
        x = np.arange(ksize, dtype=np.float64) - (ksize - 1) * 0.5
        k = np.exp(-0.5 / (s * s) * x * x)
    k = k.astype(F32)
    return (k.astype(np.float64) * (1.0 / float(k.astype(np.float64).sum()))).astype(F32)


def _reflect101(i: np.ndarray, n: int) -> np.ndarray:
    if n == 1:
        return np.zeros_like(i)
    p = 2 * (n - 1)
    i = np.abs(i) % p
    return np.where(i >= n, p - i, i)


def gaussian_blur(img: np.ndarray, ksize: int, sigma: float) -> np.ndarray:
    """cv::GaussianBlur on a CV_32F image, separable, BORDER_REFLECT_101 (rows first, then columns, float accumulation)."""
    k = gaussian_kernel(ksize, sigma)
    r = ksize // 2
    h, w = img.shape
    xs = _reflect101(np.arange(-r, w + r), w)
    tmp = np.zeros((h, w), dtype=F32)
    for j in range(ksize):
        tmp = (tmp + k[j] * img[:, xs[j:j + w]]).astype(F32)
    ys = _reflect101(np.arange(-r, h + r), h)
    out = np.zeros((h, w), dtype=F32)
    for j in range(ksize):
        out = (out + k[j] * tmp[ys[j:j + h], :]).astype(F32)
    return out


def _linear_taps(dst: int, src: int):
    """cv::resize INTER_LINEAR coordinate rule: centre-aligned, clamped at the borders."""
    scale = float(src) / float(dst)
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(F32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(F32)).astype(F32)
    lo = s < 0
    f[lo], s[lo] = 0.0, 0
    hi = s >= src - 1
    f[hi], s[hi] = 0.0, src - 1
    return s, np.minimum(s + 1, src - 1), f


def resize_linear(img: np.ndarray, width: int, height: int) -> np.ndarray:
    """cv::resize(..., INTER_LINEAR) for CV_32F images with 1 or more channels (same size = copy)."""
    h, w = img.shape[:2]
    if (h, w) == (height, width):
        return img.copy()
    x0, x1, fx = _linear_taps(width, w)
    y0, y1, fy = _linear_taps(height, h)
    a = img if img.ndim == 3 else img[:, :, None]
    fxv = fx[None, :, None]
    rows0 = (a[y0][:, x0] * (F32(1) - fxv) + a[y0][:, x1] * fxv).astype(F32)
    rows1 = (a[y1][:, x0] * (F32(1) - fxv) + a[y1][:, x1] * fxv).astype(F32)
    fyv = fy[:, None, None]
    out = (rows0 * (F32(1) - fyv) + rows1 * fyv).astype(F32)
    return out if img.ndim == 3 else out[:, :, 0]


# ---------------------------------------------------------------------------------------------------------------------
# optflowgf.cpp
# ---------------------------------------------------------------------------------------------------------------------
def prepare_gaussian(n: int = POLY_N, sigma: float = POLY_SIGMA):
    """FarnebackPrepareGaussian: applicability g, x*g, x^2*g (float) and the four entries of inv(G) that are used (double)."""
    if sigma < np.finfo(np.float32).eps:
        sigma = n * 0.3
    xs = np.arange(-n, n + 1)
    g = np.exp(-(xs * xs) / (2.0 * sigma * sigma)).astype(F32)
    s = 1.0 / float(g.astype(np.float64).sum())
    g = (g.astype(np.float64) * s).astype(F32)
    xg = (xs * g).astype(F32)
    xxg = (xs * xs * g).astype(F32)
    G = np.zeros((6, 6), dtype=np.float64)
    gd = g.astype(np.float64)
    for y in range(-n, n + 1):
        for x in range(-n, n + 1):
            wgt = gd[y + n] * gd[x + n]
            G[0, 0] += wgt
            G[1, 1] += wgt * x * x
            G[3, 3] += wgt * x * x * x * x
            G[5, 5] += wgt * x * x * y * y
    G[2, 2] = G[0, 3] = G[0, 4] = G[3, 0] = G[4, 0] = G[1, 1]
    G[4, 4] = G[3, 3]
    G[3, 4] = G[4, 3] = G[5, 5]
    inv = np.linalg.inv(G)
    return g, xg, xxg, inv[1, 1], inv[0, 3], inv[3, 3], inv[5, 5]


def poly_exp(src: np.ndarray, n: int = POLY_N, sigma: float = POLY_SIGMA) -> np.ndarray:
    """FarnebackPolyExp: [h, w] float -> [h, w, 5] float = (r3 ~ y, r2 ~ x, r5 ~ y^2, r4 ~ x^2, r6 ~ xy) coefficients; separable
    weighted least squares, replicated borders; vertical pass in float, horizontal accumulators in double."""
    g, xg, xxg, ig11, ig03, ig33, ig55 = prepare_gaussian(n, sigma)
    h, w = src.shape
    src = src.astype(F32)
    ys = np.arange(h)
    r0 = (src * g[n]).astype(F32)
    r1 = np.zeros((h, w), dtype=F32)
    r2 = np.zeros((h, w), dtype=F32)
    for k in range(1, n + 1):
        s0 = src[np.maximum(ys - k, 0)]
        s1 = src[np.minimum(ys + k, h - 1)]
        p = (s0 + s1).astype(F32)
        r0 = (r0 + g[n + k] * p).astype(F32)
        r1 = (r1 + xg[n + k] * (s1 - s0).astype(F32)).astype(F32)
        r2 = (r2 + xxg[n + k] * p).astype(F32)
    xi = np.clip(np.arange(-n, w + n), 0, w - 1)
    r0, r1, r2 = r0[:, xi], r1[:, xi], r2[:, xi]               # replicate n columns on both sides
    c = slice(n, n + w)
    b1 = (r0[:, c] * g[n]).astype(F32).astype(np.float64)
    b3 = (r1[:, c] * g[n]).astype(F32).astype(np.float64)
    b5 = (r2[:, c] * g[n]).astype(F32).astype(np.float64)
    b2 = np.zeros((h, w), dtype=np.float64)
    b4 = np.zeros((h, w), dtype=np.float64)
    b6 = np.zeros((h, w), dtype=np.float64)
    for k in range(1, n + 1):
        p, m = slice(n + k, n + k + w), slice(n - k, n - k + w)
        tg = (r0[:, p] + r0[:, m]).astype(F32).astype(np.float64)
        b1 = b1 + tg * float(g[n + k])
        b4 = b4 + tg * float(xxg[n + k])
        b2 = b2 + ((r0[:, p] - r0[:, m]).astype(F32) * xg[n + k]).astype(F32).astype(np.float64)
        b3 = b3 + ((r1[:, p] + r1[:, m]).astype(F32) * g[n + k]).astype(F32).astype(np.float64)
        b6 = b6 + ((r1[:, p] - r1[:, m]).astype(F32) * xg[n + k]).astype(F32).astype(np.float64)
        b5 = b5 + ((r2[:, p] + r2[:, m]).astype(F32) * g[n + k]).astype(F32).astype(np.float64)
    out = np.empty((h, w, 5), dtype=F32)
    out[..., 1] = (b2 * ig11).astype(F32)
    out[..., 0] = (b3 * ig11).astype(F32)
    out[..., 3] = (b1 * ig03 + b4 * ig33).astype(F32)
    out[..., 2] = (b1 * ig03 + b5 * ig33).astype(F32)
    out[..., 4] = (b6 * ig55).astype(F32)
    return out


def update_matrices(R0: np.ndarray, R1: np.ndarray, flow: np.ndarray) -> np.ndarray:
    """FarnebackUpdateMatrices: per pixel, sample R1 bilinearly at the displaced position, average the quadratic parts, form
    the 2x2 normal matrix G and right-hand side h; 5-pixel border attenuation.  All float."""
    h, w = flow.shape[:2]
    one, half, quarter = F32(1), F32(0.5), F32(0.25)
    xs = np.arange(w, dtype=F32)[None, :]
    ys = np.arange(h, dtype=F32)[:, None]
    dx, dy = flow[..., 0].astype(F32), flow[..., 1].astype(F32)
    fx = (xs + dx).astype(F32)
    fy = (ys + dy).astype(F32)
    with np.errstate(invalid="ignore"):
        x1 = np.floor(fx).astype(np.int64)
        y1 = np.floor(fy).astype(np.int64)
    fx = (fx - x1.astype(F32)).astype(F32)
    fy = (fy - y1.astype(F32)).astype(F32)
    inside = (x1 >= 0) & (x1 < w - 1) & (y1 >= 0) & (y1 < h - 1)
    xc, yc = np.where(inside, x1, 0), np.where(inside, y1, 0)
    a00 = ((one - fx) * (one - fy)).astype(F32)[..., None]
    a01 = (fx * (one - fy)).astype(F32)[..., None]
    a10 = ((one - fx) * fy).astype(F32)[..., None]
    a11 = (fx * fy).astype(F32)[..., None]
    p00, p01 = R1[yc, xc], R1[yc, np.minimum(xc + 1, w - 1)]
    p10, p11 = R1[np.minimum(yc + 1, h - 1), xc], R1[np.minimum(yc + 1, h - 1), np.minimum(xc + 1, w - 1)]
    s = (((a00 * p00).astype(F32) + (a01 * p01).astype(F32)).astype(F32) + (a10 * p10).astype(F32)).astype(F32)
    s = (s + (a11 * p11).astype(F32)).astype(F32)                                      # [h, w, 5] = r2..r6 of the moved frame
    r2 = np.where(inside, s[..., 0], F32(0))
    r3 = np.where(inside, s[..., 1], F32(0))
    r4 = np.where(inside, ((R0[..., 2] + s[..., 2]).astype(F32) * half).astype(F32), R0[..., 2])
    r5 = np.where(inside, ((R0[..., 3] + s[..., 3]).astype(F32) * half).astype(F32), R0[..., 3])
    r6 = np.where(inside, ((R0[..., 4] + s[..., 4]).astype(F32) * quarter).astype(F32), (R0[..., 4] * half).astype(F32))
    r2 = ((R0[..., 0] - r2).astype(F32) * half).astype(F32)
    r3 = ((R0[..., 1] - r3).astype(F32) * half).astype(F32)
    r2 = (r2 + ((r4 * dy).astype(F32) + (r6 * dx).astype(F32)).astype(F32)).astype(F32)   # r2 += r4*dy + r6*dx
    r3 = (r3 + ((r6 * dy).astype(F32) + (r5 * dx).astype(F32)).astype(F32)).astype(F32)

    # scale = bx(x) * bx'(x) * by(y) * by'(y), multiplied left to right as in the C++ expression; applied where the C++'s
    # unsigned range test `(unsigned)(x - BORDER) >= (unsigned)(width - BORDER*2) || (same in y)` fires
    ex_lo = np.array([BORDER[i] if i < 5 else 1.0 for i in range(w)], dtype=F32)
    ex_hi = np.array([BORDER[w - i - 1] if i >= w - 5 else 1.0 for i in range(w)], dtype=F32)
    ey_lo = np.array([BORDER[i] if i < 5 else 1.0 for i in range(h)], dtype=F32)
    ey_hi = np.array([BORDER[h - i - 1] if i >= h - 5 else 1.0 for i in range(h)], dtype=F32)
    sc = ((ex_lo * ex_hi).astype(F32)[None, :] * ey_lo[:, None]).astype(F32)
    sc = (sc * ey_hi[:, None]).astype(F32)
    u32 = lambda v: np.asarray(v, dtype=np.int64) & 0xFFFFFFFF  # noqa: E731
    edge_x = u32(np.arange(w) - 5) >= u32(w - 10)
    edge_y = u32(np.arange(h) - 5) >= u32(h - 10)
    sc = np.where(edge_x[None, :] | edge_y[:, None], sc, F32(1)).astype(F32)
    r2, r3, r4, r5, r6 = [(v * sc).astype(F32) for v in (r2, r3, r4, r5, r6)]
    M = np.empty((h, w, 5), dtype=F32)
    M[..., 0] = ((r4 * r4).astype(F32) + (r6 * r6).astype(F32)).astype(F32)
    M[..., 1] = ((r4 + r5).astype(F32) * r6).astype(F32)
    M[..., 2] = ((r5 * r5).astype(F32) + (r6 * r6).astype(F32)).astype(F32)
    M[..., 3] = ((r4 * r2).astype(F32) + (r6 * r3).astype(F32)).astype(F32)
    M[..., 4] = ((r6 * r2).astype(F32) + (r5 * r3).astype(F32)).astype(F32)
    return M


def blur_solve(M: np.ndarray, block: int = WINSIZE) -> np.ndarray:
    """FarnebackUpdateFlow_Blur (flow part): box window block x block with replicated borders (double sums), then per pixel
    flow = solve([[g11,g12],[g12,g22]] + 1e-3 regularised determinant, [h1,h2])."""
    h, w = M.shape[:2]
    m = block // 2
    Md = M.astype(np.float64)
    yi = np.clip(np.arange(-m, h + m), 0, h - 1)
    cs = np.concatenate([np.zeros((1, w, 5)), np.cumsum(Md[yi], axis=0)], axis=0)
    v = cs[block:block + h] - cs[0:h]
    xi = np.clip(np.arange(-m, w + m), 0, w - 1)
    cs = np.concatenate([np.zeros((h, 1, 5)), np.cumsum(v[:, xi], axis=1)], axis=1)
    s = (cs[:, block:block + w] - cs[:, 0:w]) * (1.0 / (block * block))
    g11, g12, g22, h1, h2 = (s[..., i] for i in range(5))
    idet = 1.0 / (g11 * g22 - g12 * g12 + 1e-3)
    flow = np.empty((h, w, 2), dtype=F32)
    flow[..., 0] = ((g11 * h2 - g12 * h1) * idet).astype(F32)
    flow[..., 1] = ((g22 * h1 - g12 * h2) * idet).astype(F32)
    return flow


def pyramid_plan(rows: int, cols: int, levels: int = LEVELS, pyr_scale: float = PYR_SCALE):
    """FarnebackOpticalFlowImpl::calc level loop -> [(scale, sigma, smooth_sz, width, height)] from the coarsest level to level 0."""
    k, scale = 0, 1.0
    while k < levels:
        scale *= pyr_scale
        if cols * scale < MIN_SIZE or rows * scale < MIN_SIZE:
            break
        k += 1
    plan = []
    for lvl in range(k, -1, -1):
        scale = 1.0
        for _ in range(lvl):
            scale *= pyr_scale
        sigma = (1.0 / scale - 1.0) * 0.5
        smooth = max(cv_round(sigma * 5) | 1, 3)
        plan.append((scale, sigma, smooth, cv_round(cols * scale), cv_round(rows * scale)))
    return plan


def calc_optical_flow_farneback(prev: np.ndarray, nxt: np.ndarray, pyr_scale: float = PYR_SCALE, levels: int = LEVELS,
                                winsize: int = WINSIZE, iterations: int = ITERS, poly_n: int = POLY_N,
                                poly_sigma: float = POLY_SIGMA) -> np.ndarray:
    """uint8 [h, w] x 2 -> flow [h, w, 2] float32 (x, y displacement), flags = 0 (box window, no initial flow)."""
    assert prev.shape == nxt.shape and prev.ndim == 2
    rows, cols = prev.shape
    imgs = [prev.astype(F32), nxt.astype(F32)]
    prev_flow = None
    for (scale, sigma, smooth, width, height) in pyramid_plan(rows, cols, levels, pyr_scale):
        if prev_flow is None:
            flow = np.zeros((height, width, 2), dtype=F32)
        else:
            flow = (resize_linear(prev_flow, width, height) * F32(1.0 / pyr_scale)).astype(F32)
        R = []
        for im in imgs:
            blurred = gaussian_blur(im, smooth, sigma)
            R.append(poly_exp(resize_linear(blurred, width, height), poly_n, poly_sigma))
        M = update_matrices(R[0], R[1], flow)
        for it in range(iterations):
            flow = blur_solve(M, winsize)
            if it < iterations - 1:
                M = update_matrices(R[0], R[1], flow)
        prev_flow = flow
    return prev_flow


# ---------------------------------------------------------------------------------------------------------------------
# the reference's wrapper around it (SCHED:156-248 as called from :382-389 / :466-476)
# ---------------------------------------------------------------------------------------------------------------------
def quantise_channel(channel: np.ndarray, gmin: np.float32, grange: np.float32) -> np.ndarray:
    """channel [T, h, w] float32 -> uint8 frames: global-range normalisation (SCHED:376-388), x255, truncation (SCHED:175-176).
    RGB2GRAY of three equal channels is the identity ((4899 + 9617 + 1868) v + 8192 >> 14 = v)."""
    norm = ((channel.astype(F32) - F32(gmin)) / F32(grange)).astype(F32)
    return (norm * F32(255)).astype(F32).astype(np.uint8)


def channel_flow(channel: np.ndarray, gmin, grange) -> np.ndarray:
    """[T, h, w] float32 -> flows [T-1, 2, h, w] float32 between consecutive frames (SCHED:199-244 without mask)."""
    q = quantise_channel(channel, gmin, grange)
    fl = [calc_optical_flow_farneback(q[t], q[t + 1]) for t in range(q.shape[0] - 1)]
    return np.stack(fl, axis=0).transpose(0, 3, 1, 2).astype(F32)
