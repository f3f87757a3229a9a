"""Oracle: depth-aware crack filling of the stage-1 warper in numpy + scipy.ndimage.  TEST INFRASTRUCTURE ONLY.

Restates /root/reference/vggt/modules/utils_warp.py `depth_aware_crack_filling` (:647-691) with everything it calls when the warper runs it
(warp_single_img :954-971, i.e. `crack_params` from create_default_crack_params and fill_segment_cracks' fast outlier test): segment_depth_map
(:506-536), fill_segment_cracks (:567-634), fill_small_cracks step 1 (:390-430; step 2 needs `depth_conf`, which fill_segment_cracks does not
pass), vectorized_depth_estimation (:539-564), merge_depth_segments (:637-676).

PARITY UNPINNED: the reference does its 3 x 3 stencils with OpenCV (opencv-python, unpinned, requirements: vggt/requirements.txt; not under
/root/reference, not in this image).  The three calls are restated from OpenCV's documented semantics:
  cv2.filter2D(src, -1, k)                       correlation, anchor at the centre, borderType BORDER_REFLECT_101 (the default)   -> ndimage.correlate(mode="mirror")
  cv2.filter2D(..., borderType=BORDER_REFLECT)   edge pixel repeated                                                              -> ndimage.correlate(mode="reflect")
  cv2.morphologyEx(m, MORPH_CLOSE, ones(3, 3))   dilate then erode, default border value = "ignore the border" for both         -> maximum_filter(cval=0) / minimum_filter(cval=1)
They are checked only against their defining properties (tests/test_oracle_crackfill.py).
"""
from __future__ import annotations

import numpy as np
from scipy import ndimage

DEFAULT_PARAMS = dict(depth_threshold=0.1, max_crack_size=5, min_valid_neighbors=3, min_neighbors=4, neighbor_radius=1)  # :694-704 (args=None)
RUN_WARP_PARAMS = dict(depth_threshold=0.1, max_crack_size=6, min_valid_neighbors=2, min_neighbors=4, neighbor_radius=1)  # run_warp.py:50-59, 294-295

K8 = np.ones((3, 3), dtype=np.float32)
K8[1, 1] = 0
K9 = np.ones((3, 3), dtype=np.float32)


def filter2d(img: np.ndarray, kernel: np.ndarray, border: str = "reflect101") -> np.ndarray:
    return ndimage.correlate(img.astype(np.float32), kernel, mode="mirror" if border == "reflect101" else "reflect")


def close3(mask_u8: np.ndarray) -> np.ndarray:
    d = ndimage.maximum_filter(mask_u8, size=3, mode="constant", cval=0)
    return ndimage.minimum_filter(d, size=3, mode="constant", cval=1)


def fill_small_cracks(image: np.ndarray, mask: np.ndarray, min_valid_neighbors: int):
    """:390-430 (step 1).  image [H,W,3] f32; mask bool / u8 -> (filled image, filled mask of mask's dtype)."""
    filled_image, filled_mask = image.copy(), mask.copy()
    if not np.any(mask == 0):
        return filled_image, filled_mask
    closed = close3(filled_mask.astype(np.uint8))
    newly = (closed > filled_mask) & (filled_mask == 0)
    if np.any(newly):
        counts = filter2d(filled_mask.astype(np.float32), K8)
        valid_fill = newly & (counts >= min_valid_neighbors)
        if np.any(valid_fill):
            safe = np.maximum(counts, 1e-6)
            mb = filled_mask > 0
            for c in range(image.shape[2]):
                s = filter2d(np.where(mb, image[:, :, c].astype(np.float32), 0.0), K8)
                filled_image[valid_fill, c] = (s / safe)[valid_fill]
            filled_mask[valid_fill] = 1
    return filled_image, filled_mask


def fill_small_cracks_depth_guided(image: np.ndarray, mask: np.ndarray, original_depth: np.ndarray, use_depth_conf: bool,
                                   depth_threshold: float = 0.1, max_crack_size: int = 5, min_valid_neighbors: int = 3):
    """:386-455 in full, as warp_single_img calls it for a view with <= 100 splatted depths (:973-981): step 1 above, then -- when a
    depth-confidence map was given (only its PRESENCE matters, :433) and step 1 filled fewer than half of the holes -- step 2: the 4-connected
    hole components (scipy.ndimage.label, the real one) of <= min(max_crack_size, 4) pixels are filled pixel by pixel, in label order and
    np.where order, from the valid 3 x 3 neighbours whose ORIGINAL depth (the source view's map, indexed at the target pixel, as the
    reference does) is within depth_threshold of the pixel's; every fill is visible to the pixels after it."""
    holes = mask == 0
    filled_image, filled_mask = fill_small_cracks(image, mask, min_valid_neighbors)
    if not np.any(holes):
        return filled_image, filled_mask
    morph_count = int(np.sum((filled_mask != 0) & holes))
    if use_depth_conf and morph_count < np.sum(holes) * 0.5:
        H, W = mask.shape
        labeled, num = ndimage.label(filled_mask == 0)
        sizes = np.bincount(labeled.ravel(), minlength=num + 1)
        for hole_id in np.nonzero((sizes <= max_crack_size) & (sizes <= 4))[0]:
            if hole_id == 0:
                continue
            ys, xs = np.where(labeled == hole_id)
            for y, x in zip(ys, xs):
                y0, y1, x0, x1 = max(0, y - 1), min(H, y + 2), max(0, x - 1), min(W, x + 2)
                valid = filled_mask[y0:y1, x0:x1] > 0
                if np.sum(valid) >= min_valid_neighbors:
                    dd = np.abs(original_depth[y0:y1, x0:x1][valid] - original_depth[y, x])
                    ok = dd <= depth_threshold
                    if np.sum(ok) >= min_valid_neighbors:
                        filled_image[y, x] = np.mean(filled_image[y0:y1, x0:x1][valid][ok], axis=0)
                        filled_mask[y, x] = 1
    return filled_image, filled_mask


def depth_estimation(depth: np.ndarray, newly: np.ndarray) -> np.ndarray:
    """:539-564."""
    if not np.any(newly):
        return depth.copy()
    ok = ~np.isnan(depth)
    dsum = filter2d(np.where(ok, depth, 0.0), K8, "reflect")
    cnt = np.maximum(filter2d(ok.astype(np.float32), K8, "reflect"), 1e-6)
    out = depth.copy()
    out[newly] = (dsum / cnt)[newly]
    return out


def segment_depth_map(depth: np.ndarray, mask: np.ndarray, num_segments: int = 5):
    """:506-536 -> list of bool masks."""
    valid = mask > 0
    vd = depth[valid]
    if len(vd) == 0:
        return []
    lo, hi = np.nanmin(vd), np.nanmax(vd)
    if lo == hi:
        return [valid]
    # the reference pins numpy < 2 (requirements.txt:6): linspace of two float32 SCALARS is float64 there (value-based promotion), float32
    # under NEP 50 -- spelled out so that this restatement follows the reference's environment on either numpy
    b = np.linspace(np.float64(lo), np.float64(hi), num_segments + 1)
    segs = []
    for i in range(num_segments):
        if i == num_segments - 1:
            segs.append((depth >= b[i]) & (depth <= b[i + 1]) & valid)
        else:
            segs.append((depth >= b[i]) & (depth < b[i + 1]) & valid)
    return segs


def fill_segment(image, depth, seg, p):
    """:567-634 with use_fast_outlier_detection (the default): returns (image, mask, depth) or None for an empty segment."""
    if np.sum(seg) == 0:
        return None
    cnt = filter2d(seg.astype(np.float32), K9)  # the fast test does NOT zero the kernel centre (:590-593)
    outlier = (seg > 0) & (cnt < p["min_neighbors"])
    cleaned = seg.copy()
    cleaned[outlier] = 0
    holes = (cleaned == 0) & (seg > 0)
    if not np.any(holes):  # a segment without outliers is returned as it is: its cracks are NOT closed (:609-611)
        return image, cleaned, depth
    fi, fm = fill_small_cracks(image, cleaned, p["min_valid_neighbors"])
    newly = (fm > 0) & (cleaned == 0)
    fd = depth_estimation(depth, newly) if np.any(newly) else depth
    return fi, fm, fd


def depth_aware_crack_filling(image: np.ndarray, mask: np.ndarray, depth: np.ndarray, params=None, num_segments: int = 5):
    """:647-691.  image [H,W,3] f32 in [0,1]; mask u8 [H,W]; depth f32 [H,W] (NaN = empty) -> (image f32, mask u8, depth f32)."""
    p = dict(DEFAULT_PARAMS)
    p.update(params or {})
    segs = segment_depth_map(depth, mask, num_segments)
    if not segs:
        fi, fm = fill_small_cracks(image, mask, p["min_valid_neighbors"])
        return fi, fm, depth
    results = [fill_segment(image, depth, s, p) for s in segs]
    H, W, C = image.shape
    mi = np.zeros((H, W, C), dtype=np.float32)
    mm = np.zeros((H, W), dtype=np.uint8)
    md = np.full((H, W), np.nan, dtype=np.float32)
    prio = []
    for i, r in enumerate(results):
        if r is not None and np.any(r[1] > 0):
            vd = r[2][~np.isnan(r[2]) & (r[1] > 0)]
            prio.append((np.mean(vd) if len(vd) > 0 else float("inf"), i, r))
    if not prio:
        fi, fm = fill_small_cracks(image, mask, p["min_valid_neighbors"])
        return fi, fm, depth
    prio.sort(key=lambda x: x[0], reverse=True)  # far to near: near segments overwrite
    for _, _, (fi, fm, fd) in prio:
        v = (fm > 0) & ~np.isnan(fd)
        if np.any(v):
            mi[v] = fi[v]
            mm[v] = fm[v]
            md[v] = fd[v]
    return mi, mm, md


def warp_frame_fill(img_u8: np.ndarray, mask_u8: np.ndarray, depth: np.ndarray, params=None, num_segments: int = 5,
                    original_depth: np.ndarray = None, use_depth_conf: bool = False):
    """The per-frame step of warp_single_img :954-985 on a splatted view: u8 image -> f32 / 255 -> fill -> (x * 255).astype(u8).
    original_depth / use_depth_conf: the source view's filtered depth map and whether a confidence map exists -- read only on the
    <= 100-splatted-depths path (:973-981)."""
    if np.sum(~np.isnan(depth)) > 100:
        fi, fm, fd = depth_aware_crack_filling(img_u8.astype(np.float32) / 255.0, mask_u8, depth, params, num_segments)
    else:
        p = dict(DEFAULT_PARAMS)
        p.update(params or {})
        od = original_depth if original_depth is not None else np.zeros(mask_u8.shape, dtype=np.float32)
        fi, fm = fill_small_cracks_depth_guided(img_u8.astype(np.float32) / 255.0, mask_u8, od, use_depth_conf and original_depth is not None,
                                                p["depth_threshold"], p["max_crack_size"], p["min_valid_neighbors"])
        fd = depth
    return (fi * 255).astype(np.uint8), fm.astype(np.uint8), fd
