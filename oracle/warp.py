"""Oracle: stage-1 depth-guided forward warping in numpy (float64).  TEST INFRASTRUCTURE ONLY.

Restates the splat of /root/reference/vggt/modules/utils_warp.py warp_single_img (:863-945) for given new cameras (the reference's
camera-path generators :64-383 are 4x4 host math and stay with the caller), without crack filling (:386-706 need OpenCV, absent from the
reference tree and from this image).  Pinned against the imported reference run with fill_cracks=False (tests/golden/g16_warp.npz,
tools/make_goldens.py warp; `import cv2` at the top of the reference module is satisfied there by an import-only placeholder).
"""
from __future__ import annotations

import numpy as np


def splat(image: np.ndarray, depth: np.ndarray, intrinsic: np.ndarray, extrinsic: np.ndarray, cameras):
    """image [H,W,3] float32 in [0,1]; depth [H,W] float32 (NaN / <= 0 invalid); intrinsic [3,3]; extrinsic [4,4] (world -> source
    camera); cameras: list of [4,4] (world -> new camera) -> (images u8 [n,H,W,3], masks u8 [n,H,W], depths f32 [n,H,W])."""
    H, W, C = image.shape
    y, x = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    pixels = np.stack([x.flatten(), y.flatten(), np.ones_like(x.flatten())], axis=-1)
    cam = np.linalg.inv(intrinsic) @ pixels.T                                   # :864
    d = depth.flatten()
    valid = ~np.isnan(d) & (d > 0)                                              # :868
    p3 = np.zeros_like(cam)
    p3[:, valid] = cam[:, valid] * d[valid]
    R, t = extrinsic[:3, :3], extrinsic[:3, 3]
    world = R.T @ p3 + (-R.T @ t)[:, None]                                      # :873-879
    xf, yf = x.flatten(), y.flatten()
    imgs, masks, depths = [], [], []
    for new_cam in cameras:
        wi = np.zeros((H, W, C), dtype=np.float32)
        wm = np.zeros((H, W), dtype=np.float32)
        wd = np.full((H, W), np.nan, dtype=np.float32)
        pc = new_cam[:3, :3] @ world + new_cam[:3, 3][:, None]                  # :900-902
        vz = (np.abs(pc[2]) > 1e-6) & valid
        if vz.sum() > 0:
            wp = np.zeros((3, pc.shape[1]))
            wp[:, vz] = intrinsic @ (pc[:, vz] / pc[2, vz])
            u, v = wp[0], wp[1]
            ok = (u >= 0) & (u < W) & (v >= 0) & (v < H) & vz
            un = np.clip(np.round(u[ok]).astype(np.int32), 0, W - 1)
            vn = np.clip(np.round(v[ok]).astype(np.int32), 0, H - 1)
            z = pc[2, ok]
            col = image[yf[ok], xf[ok]]
            order = np.argsort(-z)                                              # far to near: the last (nearest) write wins  :930-938
            wi[vn[order], un[order]] = col[order]
            wm[vn[order], un[order]] = 1.0
            wd[vn[order], un[order]] = z[order]
        imgs.append((wi * 255).astype(np.uint8) if wi.max() <= 1.0 else wi.astype(np.uint8))  # :947-950
        masks.append((wm > 0).astype(np.uint8))
        depths.append(wd)
    return np.stack(imgs), np.stack(masks), np.stack(depths)
