"""CPU: oracle/crackfill.py -- the restated OpenCV primitives against their defining properties, and the crack-filling logic of
utils_warp.py:386-691 on constructed cases (opencv-python is absent: parity with a real cv2 is unpinned)."""
import numpy as np

from oracle import crackfill as cf


def test_filter2d_is_correlation_with_reflect101_border():
    a = np.arange(20, dtype=np.float32).reshape(4, 5)
    k = np.array([[1, 2, 3], [4, 5, 6], [7, 8, 9]], dtype=np.float32)
    got = cf.filter2d(a, k)
    p = np.pad(a, 1, mode="reflect")            # numpy 'reflect' = OpenCV BORDER_REFLECT_101 (gfedcb|abcdefgh|gfedcba)
    want = np.zeros_like(a)
    for y in range(4):
        for x in range(5):
            want[y, x] = (p[y:y + 3, x:x + 3] * k).sum()
    np.testing.assert_allclose(got, want, rtol=1e-6)
    p2 = np.pad(a, 1, mode="symmetric")         # BORDER_REFLECT (fedcba|abcdefgh|hgfedcb)
    got2 = cf.filter2d(a, k, "reflect")
    assert got2[0, 0] == (p2[0:3, 0:3] * k).sum()


def test_closing_fills_one_pixel_cracks_and_ignores_the_border():
    m = np.ones((7, 9), dtype=np.uint8)
    m[:, 4] = 0                                  # a one-pixel crack, border to border
    m[3, 7] = 0                                  # a pinhole
    c = cf.close3(m)
    assert c.all()                               # border pixels of the crack are closed too: the border does not erode
    m2 = np.zeros((7, 9), dtype=np.uint8)
    m2[2:5, 2:5] = 1
    np.testing.assert_array_equal(cf.close3(m2), m2)   # closing never grows a convex blob
    wide = np.ones((7, 9), dtype=np.uint8)
    wide[:, 3:6] = 0                             # three pixels wide: not closed
    assert cf.close3(wide)[:, 4].sum() == 0


def _scene(H=24, W=32):
    """Two depth layers (near plane on the left half, far plane everywhere else), a one-pixel crack through each, one isolated far pixel
    inside a hole (an outlier) next to the far layer's crack."""
    img = np.zeros((H, W, 3), dtype=np.float32)
    img[..., 0] = np.linspace(0.2, 0.8, W)[None, :]
    img[..., 1] = np.linspace(0.1, 0.9, H)[:, None]
    img[..., 2] = 0.5
    depth = np.full((H, W), 4.0, dtype=np.float32)
    depth[:, :W // 2] = 1.0
    mask = np.ones((H, W), dtype=np.uint8)
    mask[:, 6] = 0                               # crack in the near layer
    mask[10, 20:] = 0                            # crack in the far layer (horizontal)
    mask[2:7, 24:29] = 0                         # a hole ...
    mask[4, 26] = 1                              # ... with one stray far pixel in it: an outlier of the far segment
    depth[mask == 0] = np.nan
    img[mask == 0] = 0
    return img, mask, depth


def test_segments_without_outliers_keep_their_cracks_segments_with_outliers_are_closed():
    img, mask, depth = _scene()
    fi, fm, fd = cf.depth_aware_crack_filling(img, mask, depth, cf.RUN_WARP_PARAMS)
    # the far segment has an outlier (the stray pixel): it is dropped, and the segment's cracks are closed
    assert fm[4, 26] == 0
    assert fm[10, 22:31].all() and not np.isnan(fd[10, 22:31]).any()
    np.testing.assert_allclose(fi[10, 25], (img[9, 24:27].sum(0) + img[11, 24:27].sum(0)) / 6, rtol=1e-5)   # mean of the valid 8-neighbours
    np.testing.assert_allclose(fd[10, 25], 4.0)
    # the near segment has no outlier: fill_segment_cracks returns it untouched (utils_warp.py:609-611) -- its crack stays open
    assert fm[:, 6].sum() == 0
    # the 5 x 5 hole is too wide for a 3 x 3 closing
    assert fm[3:6, 25:28].sum() == 0
    # everything else is carried over
    keep = (mask > 0)
    keep[4, 26] = False
    np.testing.assert_array_equal(fi[keep], img[keep])


def test_thin_near_structures_are_outliers_of_their_segment_and_get_painted_over():
    """A one-pixel-thin near object on a far plane: every one of its pixels has < 4 same-segment pixels in its 3 x 3 window, so the near
    segment drops them all (fast outlier test, utils_warp.py:586-600) -- and the far segment, having an outlier of its own elsewhere, closes
    the resulting one-pixel gap with the mean of the far neighbours.  (What the reference does, restated; not a judgement.)"""
    H, W = 16, 16
    img = np.full((H, W, 3), 0.25, dtype=np.float32)
    depth = np.full((H, W), 5.0, dtype=np.float32)
    mask = np.ones((H, W), dtype=np.uint8)
    depth[8, 4:12] = 1.0
    img[8, 4:12] = 0.9
    mask[1:4, 1:4] = 0
    mask[2, 2] = 1                               # stray far pixel -> the far segment has an outlier -> its mask gets closed
    depth[mask == 0] = np.nan
    fi, fm, fd = cf.depth_aware_crack_filling(img, mask, depth, cf.RUN_WARP_PARAMS)
    np.testing.assert_allclose(fi[8, 5:11], 0.25)
    # vectorized_depth_estimation averages over every non-NaN depth around the pixel, the dropped outliers' included: (6 x 5 + 2 x 1) / 8
    np.testing.assert_allclose(fd[8, 5:11], 4.0)
    assert fm[8, 4:12].all() and fm[2, 2] == 0


def test_two_pixel_thick_near_structures_survive():
    H, W = 16, 16
    img = np.full((H, W, 3), 0.25, dtype=np.float32)
    depth = np.full((H, W), 5.0, dtype=np.float32)
    mask = np.ones((H, W), dtype=np.uint8)
    depth[8:10, 4:12] = 1.0
    img[8:10, 4:12] = 0.9
    mask[1:4, 1:4] = 0
    mask[2, 2] = 1
    depth[mask == 0] = np.nan
    fi, fm, fd = cf.depth_aware_crack_filling(img, mask, depth, cf.RUN_WARP_PARAMS)
    np.testing.assert_allclose(fi[8:10, 4:12], 0.9)
    np.testing.assert_allclose(fd[8:10, 4:12], 1.0)


def test_fill_small_cracks_with_depth_guided_step_equals_reference():
    """G23: the reference's own fill_small_cracks (utils_warp.py:386-455, imported unmodified; its two OpenCV stencils served by this
    oracle's restatements, everything else -- ndimage.label, the size rules, the sequential depth-guided fill -- the reference's code) on a
    <= 100-pixel view, 8 parameter sets incl. ones where step 2 fills pixels."""
    import os
    from tests.cases import SMALL_CRACK_CASES
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "g23_fill_small_cracks.npz"))
    img, mask, depth = g["img"], g["mask"], g["depth"]
    assert int(mask.sum()) <= 100
    step2 = 0
    for name, (has_conf, mcs, mvn, thr) in SMALL_CRACK_CASES.items():
        fi, fm = cf.fill_small_cracks_depth_guided(img, mask, depth, has_conf, thr, mcs, mvn)
        assert np.array_equal(fm, g[f"{name}_mask"]), name
        np.testing.assert_array_equal(fi, g[f"{name}_img"], err_msg=name)
        step2 += int(fm.sum() - cf.fill_small_cracks(img, mask, mvn)[1].sum())
    assert step2 >= 5   # the depth-guided step did fill pixels in some of the cases
