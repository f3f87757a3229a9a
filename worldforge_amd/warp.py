"""Stage-1 depth-guided forward warping on the GPU (vggt/modules/utils_warp.py warp_single_img :863-945, without crack filling).

`forward_splat` is the tensor-sized part of the stage-1 warper: one image + depth map -> n warped views with validity masks (the
`warp_*` / `mask_*` frames the guided sampler consumes).  The camera paths (:64-383) are 4x4 host math and the confidence filter a
percentile on the host: both stay with the caller, as does the OpenCV-based crack filling (:386-706, not built).
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops
from ._ffi import call


def forward_splat(image: torch.Tensor, depth: torch.Tensor, intrinsic, extrinsic, cameras, device="cuda:0"):
    """image [H,W,3] float in [0,1]; depth [H,W] float (NaN / <= 0 = invalid); intrinsic 3x3; extrinsic 4x4 or 3x4 (world -> source
    camera); cameras: sequence of 4x4 (world -> new camera).  Returns (images u8 [n,H,W,3], masks u8 [n,H,W], depths f32 [n,H,W]) on the
    device."""
    dev = torch.device(device)
    img = torch.as_tensor(image, dtype=torch.float32).to(dev).contiguous()
    dep = torch.as_tensor(depth, dtype=torch.float32).to(dev).contiguous()
    H, W, C = img.shape
    if C != 3 or tuple(dep.shape) != (H, W):
        raise ValueError("image must be [H, W, 3] and depth [H, W]")
    K = np.asarray(intrinsic, dtype=np.float64)
    E = np.eye(4)
    ext = np.asarray(extrinsic, dtype=np.float64)
    E[:ext.shape[0], :] = ext
    R, t = E[:3, :3], E[:3, 3]
    geom = np.concatenate([np.linalg.inv(K).ravel(), K.ravel(), R.T.ravel(), (-R.T @ t).ravel()])
    cams = np.stack([np.asarray(c, dtype=np.float64)[:3, :4].ravel() for c in cameras])
    n = cams.shape[0]
    geom_d = torch.from_numpy(geom).to(dev)
    cams_d = torch.from_numpy(cams).contiguous().to(dev)
    out_img = torch.empty((n, H, W, 3), dtype=torch.uint8, device=dev)
    out_mask = torch.empty((n, H, W), dtype=torch.uint8, device=dev)
    out_depth = torch.empty((n, H, W), dtype=torch.float32, device=dev)
    zbuf = torch.empty((n, H, W), dtype=torch.int64, device=dev)
    call("wf_warp_splat", img.data_ptr(), dep.data_ptr(), geom_d.data_ptr(), cams_d.data_ptr(), out_img.data_ptr(), out_mask.data_ptr(),
         out_depth.data_ptr(), zbuf.data_ptr(), n, H, W, ops.stream())
    return out_img, out_mask, out_depth
