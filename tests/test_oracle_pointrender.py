"""CPU: oracle/pointrender.py (the DepthCrafter warper's point renderer; pytorch3d + OpenCV restated, parity UNPINNED) against the defining
properties of what it restates, and scipy's own filters where the reference calls scipy."""
import numpy as np
from scipy import ndimage

from oracle import pointrender as pr


def _scene(H=320, W=512, seed=0):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    disp = (0.3 + 0.2 * np.sin(xx / 40.0) + 0.25 * (yy > H // 2) + 0.15 * ((xx // 64) % 2)).astype(np.float32)
    depth = (1.0 / (disp + 0.1)).astype(np.float32)
    rgb = rng.random((H, W, 3)).astype(np.float32)
    K = np.array([[525, 0, W / 2], [0, 525, H / 2], [0, 0, 1]], dtype=np.float32)
    return rgb, depth, K


def test_pixel_centres_follow_the_ndc_convention():
    # short side spans [-1, 1], long side [-W/H, W/H]; centres half a pixel inside; index 0 is the LEFT-most NDC value after the flip
    H, W = 480, 832
    xs = pr.pix_to_ndc(np.arange(W), W, H)
    ys = pr.pix_to_ndc(np.arange(H), H, W)
    assert np.isclose(xs[0], -W / H + 1.0 / H) and np.isclose(xs[-1], W / H - 1.0 / H)
    assert np.isclose(ys[0], -1 + 1.0 / H) and np.isclose(ys[-1], 1 - 1.0 / H)
    assert np.allclose(np.diff(xs), 2.0 / H, atol=1e-6) and np.allclose(np.diff(ys), 2.0 / H, atol=1e-6)


def test_identity_camera_sees_every_pixel_through_its_own_or_next_point():
    """K's principal point is (W/2, H/2) and pixel centres sit at half-integers in pytorch3d: the point of pixel (y, x) lands on the corner
    shared by pixels (y-1..y, x-1..x); with radius 0.005 * H/2 > sqrt(0.5) px each pixel is covered by the points of (y..y+1, x..x+1) and
    shows the nearest of them."""
    rgb, depth, K = _scene()
    H, W = depth.shape
    pts = pr.unproject(depth, K)
    x, y, z = pr.to_ndc(pts, *pr.cameras_from_opencv(np.eye(4), K, (H, W)))
    idx = pr.rasterize_nearest(x, y, z, (H, W))
    assert (idx >= 0).all()
    off = idx - np.arange(H * W).reshape(H, W)
    assert set(np.unique(off)) <= {0, 1, W, W + 1}
    cand = np.stack([depth, np.roll(depth, -1, 1), np.roll(depth, -1, 0), np.roll(np.roll(depth, -1, 0), -1, 1)])
    inner = (slice(0, H - 1), slice(0, W - 1))
    assert np.array_equal(pts[idx[inner], 2], cand[(slice(None),) + inner].min(0))


def test_points_behind_the_camera_and_tiny_images_render_nothing():
    rgb, depth, K = _scene(64, 96)
    cam = np.eye(4)
    img, mask = pr.render_frame(rgb, depth, cam, K, False)     # radius 0.005 * 32 = 0.16 px < half a pixel: no centre is reached
    assert mask.sum() == 0 and not img.any()
    rgb, depth, K = _scene()
    back = np.eye(4)
    back[2, 3] = -100.0                                          # every point behind the camera
    img, mask = pr.render_frame(rgb, depth, back, K, False)
    assert mask.sum() == 0 and not img.any()


def test_opening_removes_specks_and_keeps_blocks():
    m = np.zeros((40, 40), np.uint8)
    m[5:15, 5:15] = 1       # 10 x 10 block survives a 5 x 5 opening unchanged
    m[30, 30] = 1           # speck
    m[20:24, 20:40] = 1     # 4 rows thick: removed, even where it touches the border (the border does not help an erosion... it does in
    o = pr.morph_open5(m)   # OpenCV: outside pixels are ignored, so only the missing rows matter)
    assert np.array_equal(o[5:15, 5:15], m[5:15, 5:15]) and o[30, 30] == 0 and not o[20:24].any()
    full = np.ones((12, 12), np.uint8)
    assert np.array_equal(pr.morph_open5(full), full)            # ignoring the border keeps a full mask full
    assert (o <= m).all()


def test_edge_filter_uses_scipys_own_filters_and_marks_depth_steps():
    _, depth, _ = _scene()
    drop = pr.edge_filter_mask(depth)
    var = ndimage.maximum_filter(depth, size=5) - ndimage.minimum_filter(depth, size=5)
    assert (drop | ~(var > 0.3)).all()                           # every depth jump is dropped
    H, W = depth.shape
    assert drop[H // 2 - 2:H // 2 + 2].all()                     # the horizontal depth step
    flat = np.full((32, 32), 2.0, np.float32)
    assert not pr.edge_filter_mask(flat).any()                   # no gradient, magnitude maximum 0: nothing normalised, nothing dropped
    ramp = np.tile(np.linspace(1, 2, 64, dtype=np.float32), (32, 1))
    e = pr.detect_depth_edges(ramp)
    assert e[:, 1:-1].all()                                      # constant gradient == its own maximum > 0.1 (REFLECT_101 zeroes the border columns)
    assert not e[:, 0].any() and not e[:, -1].any()


def test_moving_the_camera_opens_disocclusions_on_the_right_side():
    rgb, depth, K = _scene()
    cam = np.eye(4)
    cam[0, 3] = 0.08        # tvec +x: the scene shifts towards +x in the image, nearer (shallower) points move further
    img, mask = pr.render_frame(rgb, depth, cam, K, True)
    assert 0.5 < mask.mean() < 1.0
    assert not img[mask[..., 0] == 0].any()
    assert mask[:, :8].mean() < mask[:, -8:].mean()              # the left border is uncovered
