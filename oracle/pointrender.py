"""Oracle: the point-cloud renderer of the DepthCrafter stage-1 warper, in numpy / scipy.  TEST INFRASTRUCTURE ONLY.

Restates /root/reference/DepthCrafter/utils.py project_points_to_image_pytorch (:103-171), detect_depth_edges (:495-520),
filter_edge_points (:523-567), project_points_to_image_pytorch_with_edge_filter (:570-599) as warp_depthcrafter.py:255-288 calls them per
frame (point cloud = the frame un-projected through K = [[525, 0, W/2], [0, 525, H/2], [0, 0, 1]] at depth 1 / (disparity + 0.1)).

PARITY UNPINNED.  The rasteriser is pytorch3d (third party, unpinned in DepthCrafter/requirements, absent from /root/reference and from
this image) and the morphology / Sobel / dilation are OpenCV (absent too).  Restated from their published behaviour:
  pytorch3d.utils.camera_conversions._cameras_from_opencv_projection: R' = R^T with its first two columns negated, T' = (-tx, -ty, tz),
      focal' = focal / s, principal' = -(principal - (W, H) / 2) / s with s = min(W, H) / 2  (NDC: +x left, +y up, short side = [-1, 1])
  PointsRasterizer.transform: view = p R' + T'; ndc_xy = (focal' * view_xy + principal' * view_z) / view_z (homogeneous divide of the
      projection matrix product); depth = view_z
  rasterize_points (naive and coarse-to-fine agree unless a bin overflows): pixel (yi, xi) has centre
      PixToNonSquareNdc(S - 1 - i, S, S_other) = -o + (r * (S - 1 - i) + o) / S with r = 2 (S <= S_other) or 2 S / S_other, o = r / 2;
      a point with view_z >= 0 covers the pixel if dx^2 + dy^2 < radius^2; fragments.idx[..., 0] = the covering point of smallest view_z
  cv2.morphologyEx(MORPH_OPEN, ones(5, 5)): erosion then dilation, anchor at the centre, the border ignored by both (erode: +inf, dilate: -inf)
  cv2.Sobel(depth, CV_64F, 1, 0, ksize=3): correlation with [[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]] (dy: its transpose), BORDER_REFLECT_101
  cv2.dilate(mask, ones(7, 7)): 7 x 7 maximum, border ignored
scipy.ndimage.minimum_filter / maximum_filter (:548-549) are called as the reference calls them (scipy is in the image: that step is exact).
"""
from __future__ import annotations

import numpy as np
from scipy import ndimage

F32 = np.float32


def cameras_from_opencv(extrinsic: np.ndarray, K: np.ndarray, size_hw):
    """-> (R' [3,3], T' [3], focal' [2], principal' [2]) float32, pytorch3d convention (row vectors: view = p @ R' + T')."""
    H, W = int(size_hw[0]), int(size_hw[1])
    R = np.asarray(extrinsic, dtype=np.float64)[:3, :3].astype(F32)
    t = np.asarray(extrinsic, dtype=np.float64)[:3, 3].astype(F32)
    K = np.asarray(K, dtype=F32)
    wh = np.array([W, H], dtype=F32)
    s = F32(min(W, H)) / F32(2.0)
    focal = np.array([K[0, 0], K[1, 1]], dtype=F32) / s
    p0 = -(np.array([K[0, 2], K[1, 2]], dtype=F32) - wh / F32(2.0)) / s
    Rp = R.T.copy()
    Rp[:, :2] *= F32(-1)
    Tp = t.copy()
    Tp[:2] *= F32(-1)
    return Rp, Tp, focal.astype(F32), p0.astype(F32)


def to_ndc(points: np.ndarray, Rp, Tp, focal, p0):
    """points [N, 3] f32 (world) -> (x_ndc, y_ndc, view_z) float32, op order as documented in the header."""
    p = points.astype(F32)
    v = np.empty_like(p)
    for j in range(3):
        v[:, j] = ((p[:, 0] * Rp[0, j] + p[:, 1] * Rp[1, j]) + p[:, 2] * Rp[2, j]) + Tp[j]
    z = v[:, 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        x = (focal[0] * v[:, 0] + p0[0] * z) / z
        y = (focal[1] * v[:, 1] + p0[1] * z) / z
    return x.astype(F32), y.astype(F32), z.astype(F32)


def pix_to_ndc(i, S1: int, S2: int):
    r = F32(2.0) if S1 <= S2 else (F32(S1) * F32(2.0)) / F32(S2)
    o = r / F32(2.0)
    return (-o + (r * np.asarray(i, dtype=F32) + o) / F32(S1)).astype(F32)


def rasterize_nearest(x, y, z, size_hw, radius: float = 0.005) -> np.ndarray:
    """-> idx int64 [H, W]: the covering point of smallest view_z per pixel, -1 = none (ties: the smaller point index)."""
    H, W = int(size_hw[0]), int(size_hw[1])
    r2 = F32(radius) * F32(radius)
    best_i = np.full((H, W), -1, dtype=np.int64)
    ok = (z >= 0) & np.isfinite(x) & np.isfinite(y)
    ids = np.nonzero(ok)[0]
    rx = 2.0 if W <= H else 2.0 * W / H
    ry = 2.0 if H <= W else 2.0 * H / W
    cx = (x[ids].astype(np.float64) + rx / 2) * W / rx - 0.5  # flipped column index i' = W - 1 - xi of the nearest centre
    cy = (y[ids].astype(np.float64) + ry / 2) * H / ry - 0.5
    reach = int(np.ceil(radius * max(W / rx, H / ry))) + 1
    bx, by = np.floor(cx).astype(np.int64), np.floor(cy).astype(np.int64)
    hp, hz, hi = [], [], []
    for dy in range(-reach, reach + 2):
        for dx in range(-reach, reach + 2):
            ix, iy = bx + dx, by + dy
            k = np.nonzero((ix >= 0) & (ix < W) & (iy >= 0) & (iy < H))[0]
            if k.size == 0:
                continue
            xf, yf = pix_to_ndc(ix[k], W, H), pix_to_ndc(iy[k], H, W)
            ddx, ddy = xf - x[ids[k]], yf - y[ids[k]]
            k = k[(ddx * ddx + ddy * ddy).astype(F32) < r2]
            hp.append((H - 1 - iy[k]) * W + (W - 1 - ix[k]))
            hz.append(z[ids[k]])
            hi.append(ids[k])
    if hp:
        pix, zz, ii = np.concatenate(hp), np.concatenate(hz), np.concatenate(hi)
        order = np.lexsort((ii, zz, pix))  # per pixel: nearest view_z first, then the smaller point index
        ps = pix[order]
        first = np.r_[True, ps[1:] != ps[:-1]] if ps.size else np.zeros(0, dtype=bool)
        best_i.reshape(-1)[ps[first]] = ii[order][first]
    return best_i


def morph_open5(mask_u8: np.ndarray) -> np.ndarray:
    e = ndimage.minimum_filter(mask_u8, size=5, mode="constant", cval=1)
    return ndimage.maximum_filter(e, size=5, mode="constant", cval=0)


def project_points_to_image(points, features, extrinsic, K, size_hw, morph: bool = True, radius: float = 0.005):
    """:103-171 -> (image f32 [H, W, F], mask u8 [H, W, 1])."""
    Rp, Tp, focal, p0 = cameras_from_opencv(extrinsic, K, size_hw)
    x, y, z = to_ndc(points, Rp, Tp, focal, p0)
    idx = rasterize_nearest(x, y, z, size_hw, radius)
    feats = np.asarray(features, dtype=F32)
    image = feats[idx]  # (-1 picks the last point, as in the reference; those pixels are zeroed below)
    mask = np.where(idx == -1, 0, 1).astype(np.uint8)
    if morph:
        mask = morph_open5(mask)
    image[mask == 0] = 0
    return image, mask[..., None]


SOBEL_X = np.array([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]], dtype=np.float64)


def detect_depth_edges(depth: np.ndarray, edge_threshold: float = 0.1) -> np.ndarray:
    """:495-520."""
    d = depth.astype(np.float64)
    gx = ndimage.correlate(d, SOBEL_X, mode="mirror")
    gy = ndimage.correlate(d, SOBEL_X.T, mode="mirror")
    mag = np.sqrt(gx ** 2 + gy ** 2)
    if mag.max() > 0:
        mag = mag / mag.max()
    return mag > edge_threshold


def edge_filter_mask(depth: np.ndarray, edge_threshold=0.1, edge_dilation=3, depth_jump_threshold=0.3, neighbor_check_radius=2) -> np.ndarray:
    """:523-560 -> bool [H, W], True = the pixel's point is DROPPED."""
    edge = detect_depth_edges(depth, edge_threshold)
    if edge_dilation > 0:
        edge = ndimage.maximum_filter(edge.astype(np.uint8), size=2 * edge_dilation + 1, mode="constant", cval=0).astype(bool)
    if depth_jump_threshold > 0 and neighbor_check_radius > 0:
        s = 2 * neighbor_check_radius + 1
        var = ndimage.maximum_filter(depth, size=s) - ndimage.minimum_filter(depth, size=s)
        edge = edge | (var > depth_jump_threshold)
    return edge


def unproject(depth_frame: np.ndarray, K: np.ndarray) -> np.ndarray:
    """warp_depthcrafter.py:259-264: [H, W] depth -> points [H * W, 3] float32."""
    H, W = depth_frame.shape
    ii, jj = np.indices((H, W))
    K = np.asarray(K, dtype=F32)
    d = depth_frame.astype(F32)
    X = (jj.astype(F32) - K[0, 2]) * d / K[0, 0]
    Y = (ii.astype(F32) - K[1, 2]) * d / K[1, 1]
    return np.stack((X, Y, d), axis=-1).reshape(-1, 3).astype(F32)


def render_frame(rgb: np.ndarray, depth_frame: np.ndarray, cam: np.ndarray, K: np.ndarray, edge_filter: bool, **kw):
    """One iteration of warp_depthcrafter.py:255-288: rgb f32 [H, W, 3], depth f32 [H, W] -> (image, mask)."""
    H, W = depth_frame.shape
    pts, feats = unproject(depth_frame, K), rgb.reshape(-1, 3).astype(F32)
    if edge_filter:
        keep = ~edge_filter_mask(depth_frame, **kw).reshape(-1)
        pts, feats = pts[keep], feats[keep]
    return project_points_to_image(pts, feats, cam, K, (H, W), morph=True)
