"""Stage-1 depth-guided forward warping on the GPU (vggt/modules/utils_warp.py warp_single_img :724-1001).

`forward_splat` (:863-945) and `crack_fill` (:954-985 -> :386-706) are the tensor-sized parts of the stage-1 warper: one image + depth map
-> n warped views with validity masks (the `warp_*` / `mask_*` frames the guided sampler consumes), cracks closed.  The camera paths
(:64-383) are 4x4 host math (`camera_path`) and the confidence filter a percentile on the host: both stay with the caller.
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


# ---- camera paths (vggt/modules/utils_warp.py:64-383 + the dispatch of warp_single_img :818-839): 4x4 host math, float64 ------------
def _rot(axis: str, rad: float) -> np.ndarray:
    c, s = np.cos(rad), np.sin(rad)
    if axis == "x":
        return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def _look_at(R0: np.ndarray, cam_pos: np.ndarray, target: np.ndarray, guarded: bool):
    """Camera at cam_pos looking at target, up vector = the source camera's y axis made orthogonal to the view direction."""
    z = target - cam_pos
    n = np.linalg.norm(z)
    if guarded and not n > 1e-6:
        return None
    z = z / n
    y0 = R0.T @ np.array([0, 1, 0])
    y = y0 - np.dot(y0, z) * z
    yn = np.linalg.norm(y)
    if guarded and not yn > 1e-6:
        y = np.array([0, 1, 0]) if abs(z[1]) < 0.9 else np.array([1, 0, 0])
        y = y - np.dot(y, z) * z
        yn = np.linalg.norm(y)
    y = y / yn
    x = np.cross(y, z)
    x = x / np.linalg.norm(x)
    return np.column_stack([x, y, z]).T


def camera_path(direction: str, extrinsic, degree: float, frame_num: int, look_at_depth: float):
    """The reference's camera sequences: `up` / `down` / `left` / `right` orbit the point `look_at_depth` in front of the camera by up
    to +-degree about the x / y axis and re-aim at it; `forward` / `backward` dolly along the view direction by degree % of that depth;
    `*_pan` rotate in place.  Returns frame_num 4x4 world -> camera matrices, the first being the source camera."""
    E = np.eye(4)
    ext = np.asarray(extrinsic, dtype=np.float64)
    E[:ext.shape[0], :] = ext
    R, t = E[:3, :3], E[:3, 3]
    cam_pos = -R.T @ t
    d = direction.lower()
    out = []
    if d in ("up", "down", "left", "right"):
        axis = "x" if d in ("up", "down") else "y"
        sign = 1.0 if d in ("up", "right") else -1.0
        look_at = cam_pos + R.T @ np.array([0, 0, look_at_depth])
        for deg in np.linspace(0, sign * degree, frame_num):
            new_pos = look_at - _rot(axis, np.deg2rad(deg)) @ (look_at - cam_pos)
            nR = _look_at(R, new_pos, look_at, guarded=False)
            out.append((nR, new_pos))
    elif d in ("forward", "backward"):
        centre = cam_pos + R.T @ np.array([0, 0, look_at_depth])
        to_c = centre - cam_pos
        radius = np.linalg.norm(to_c)
        step = (to_c / radius) if d == "forward" else (-to_c / radius)
        for prog in np.linspace(0, degree / 100.0, frame_num):
            new_pos = cam_pos + step * (radius * prog)
            nR = _look_at(R, new_pos, centre, guarded=True)
            out.append((R.copy() if nR is None else nR, new_pos))
    elif d in ("up_pan", "down_pan", "left_pan", "right_pan"):
        axis = "x" if d in ("up_pan", "down_pan") else "y"
        sign = 1.0 if d in ("up_pan", "right_pan") else -1.0
        for deg in np.linspace(0, degree, frame_num):
            out.append((R @ _rot(axis, np.deg2rad(sign * deg)), cam_pos))
    else:
        raise ValueError(f"Unsupported direction: {direction}")
    cams = []
    for nR, pos in out:
        c = E.copy()
        c[:3, :3] = nR
        c[:3, 3] = -nR @ pos
        cams.append(c)
    return cams


def crack_fill(images: torch.Tensor, masks: torch.Tensor, depths: torch.Tensor, min_neighbors: int = 4, min_valid_neighbors: int = 3,
               num_segments: int = 5, original_depth: torch.Tensor = None, has_depth_conf: bool = False, depth_threshold: float = 0.1,
               max_crack_size: int = 5):
    """Crack filling of splatted views as warp_single_img applies it per view (utils_warp.py:954-985): images u8 [n,H,W,3], masks u8
    [n,H,W], depths f32 [n,H,W] (NaN = empty), as forward_splat returns them -> the same three, filled.  Views with more than 100 splatted
    depths take depth_aware_crack_filling (:647-691); the others fill_small_cracks (:386-455, `small_crack_fill` below) with the SOURCE
    view's filtered depth map `original_depth` [H,W] and `has_depth_conf` (whether the caller has a confidence map: only its presence
    matters, :433) -- their depth is returned unchanged, as in the reference.  Defaults = create_default_crack_params(None) (:694-704);
    run_warp.py passes min_valid_neighbors = 2, max_crack_size = 6 (:50-59, 294)."""
    n, H, W, _ = images.shape
    assert images.dtype == torch.uint8 and masks.dtype == torch.uint8 and depths.dtype == torch.float32
    images, masks, depths = images.contiguous(), masks.contiguous(), depths.contiguous()
    from ._ffi import lib
    ws = torch.empty(int(lib().wf_crack_fill_workspace_bytes(n, H, W)), dtype=torch.uint8, device=images.device)
    oi, om, od = torch.empty_like(images), torch.empty_like(masks), torch.empty_like(depths)
    call("wf_crack_fill", images.data_ptr(), masks.data_ptr(), depths.data_ptr(), oi.data_ptr(), om.data_ptr(), od.data_ptr(), n, H, W,
         int(min_neighbors), int(min_valid_neighbors), int(num_segments), ws.data_ptr(), ops.stream())
    few = (~torch.isnan(depths)).flatten(1).sum(1) <= 100
    if bool(few.any()):  # once per camera path, outside any loop
        for i in torch.nonzero(few).flatten().tolist():
            oi[i], om[i] = small_crack_fill(images[i], masks[i], original_depth, has_depth_conf, depth_threshold, max_crack_size,
                                            min_valid_neighbors)
            od[i] = depths[i]
    return oi, om, od


def small_crack_fill(image: torch.Tensor, mask: torch.Tensor, original_depth: torch.Tensor = None, has_depth_conf: bool = False,
                     depth_threshold: float = 0.1, max_crack_size: int = 5, min_valid_neighbors: int = 3):
    """fill_small_cracks (utils_warp.py:386-455) of ONE view: image u8 [H,W,3], mask u8 [H,W]; original_depth f32 [H,W] (the source view's
    filtered depth) is read by the depth-guided second step, which runs only when a confidence map exists (has_depth_conf)."""
    from ._ffi import lib
    H, W, _ = image.shape
    if has_depth_conf and original_depth is None:
        raise ValueError("small_crack_fill: has_depth_conf needs original_depth (the source view's filtered depth map)")
    image, mask = image.contiguous(), mask.contiguous()
    od = original_depth.to(torch.float32).contiguous() if original_depth is not None else None
    ws = torch.empty(int(lib().wf_fill_small_cracks_workspace_bytes(H, W)), dtype=torch.uint8, device=image.device)
    oi, om = torch.empty_like(image), torch.empty_like(mask)
    call("wf_fill_small_cracks", image.data_ptr(), mask.data_ptr(), od.data_ptr() if od is not None else None, 1 if has_depth_conf else 0,
         oi.data_ptr(), om.data_ptr(), H, W, float(depth_threshold), int(max_crack_size), int(min_valid_neighbors), ws.data_ptr(), ops.stream())
    return oi, om


# ---- dynamic scenes: the DepthCrafter warper's per-frame point-cloud render (DepthCrafter/warp_depthcrafter.py:255-288) ------------------
def pytorch3d_camera(extrinsic, K, size_hw) -> np.ndarray:
    """The 16 floats wf_points_render takes: pytorch3d's view of the reference's (extrinsic, K, image size), i.e. what
    pytorch3d.utils.camera_conversions._cameras_from_opencv_projection builds at DepthCrafter/utils.py:119-124 -- R' = R^T with its first two
    columns negated, T' = (-tx, -ty, tz), focal / s, -(principal - (W, H) / 2) / s with s = min(W, H) / 2, all float32."""
    H, W = int(size_hw[0]), int(size_hw[1])
    E = np.asarray(extrinsic, dtype=np.float64)
    R, t = E[:3, :3].astype(np.float32), E[:3, 3].astype(np.float32)
    Kf = np.asarray(K, dtype=np.float32)
    s = np.float32(min(W, H)) / np.float32(2.0)
    Rp = R.T.copy()
    Rp[:, :2] *= np.float32(-1)
    Tp = t.copy()
    Tp[:2] *= np.float32(-1)
    focal = np.array([Kf[0, 0], Kf[1, 1]], dtype=np.float32) / s
    p0 = -(np.array([Kf[0, 2], Kf[1, 2]], dtype=np.float32) - np.array([W, H], dtype=np.float32) / np.float32(2.0)) / s
    return np.concatenate([Rp.reshape(-1), Tp, focal, p0]).astype(np.float32)


def depth_edge_mask(depth: torch.Tensor, edge_threshold: float = 0.1, edge_dilation: int = 3, depth_jump_threshold: float = 0.3,
                    neighbor_check_radius: int = 2) -> torch.Tensor:
    """filter_edge_points (DepthCrafter/utils.py:523-567): depth f32 [H, W] (device) -> u8 [H, W], 1 = the pixel's point is dropped."""
    from ._ffi import lib
    H, W = depth.shape
    depth = depth.to(torch.float32).contiguous()
    ws = torch.empty(int(lib().wf_depth_edge_mask_workspace_bytes(H, W)), dtype=torch.uint8, device=depth.device)
    out = torch.empty((H, W), dtype=torch.uint8, device=depth.device)
    call("wf_depth_edge_mask", depth.data_ptr(), H, W, float(edge_threshold), int(edge_dilation), float(depth_jump_threshold),
         int(neighbor_check_radius), out.data_ptr(), ws.data_ptr(), ops.stream())
    return out


def points_render(points: torch.Tensor, features: torch.Tensor, extrinsic, K, size_hw, morph: bool = True, radius: float = 0.005,
                  drop: torch.Tensor = None):
    """project_points_to_image_pytorch (DepthCrafter/utils.py:103-171): points f32 [n, 3], features f32 [n, F] (device), extrinsic 4x4 and K
    3x3 (host), -> (image f32 [H, W, F], mask u8 [H, W, 1]) on the device.  drop: u8 [n], 1 = point removed beforehand (edge filter)."""
    from ._ffi import lib
    H, W = int(size_hw[0]), int(size_hw[1])
    points, features = points.to(torch.float32).contiguous(), features.to(torch.float32).contiguous()
    n, F = features.shape
    assert tuple(points.shape) == (n, 3)
    cam = pytorch3d_camera(extrinsic, K, (H, W))
    ws = torch.empty(int(lib().wf_points_render_workspace_bytes(H, W)), dtype=torch.uint8, device=points.device)
    img = torch.empty((H, W, F), dtype=torch.float32, device=points.device)
    mask = torch.empty((H, W), dtype=torch.uint8, device=points.device)
    call("wf_points_render", points.data_ptr(), features.data_ptr(), drop.contiguous().data_ptr() if drop is not None else None, n, F,
         cam.ctypes.data, H, W, float(radius), 1 if morph else 0, img.data_ptr(), mask.data_ptr(), ws.data_ptr(), ops.stream())
    return img, mask.unsqueeze(-1)


def render_depthcrafter_frame(rgb: torch.Tensor, depth_frame: torch.Tensor, cam, K, edge_filter: bool = True, **edge_kw):
    """One iteration of warp_depthcrafter.py:255-288: rgb f32 [H, W, 3] in [0, 1], depth_frame f32 [H, W] (= 1 / (disparity + 0.1)),
    cam 4x4, K 3x3 -> (rendered image f32 [H, W, 3], mask u8 [H, W, 1]); the reference skips the edge filter for frame 0."""
    H, W = depth_frame.shape
    Kf = np.asarray(K, dtype=np.float32)
    d = depth_frame.to(torch.float32)
    ii = torch.arange(H, device=d.device, dtype=torch.float32).view(H, 1)
    jj = torch.arange(W, device=d.device, dtype=torch.float32).view(1, W)
    pts = torch.stack(((jj - float(Kf[0, 2])) * d / float(Kf[0, 0]), (ii - float(Kf[1, 2])) * d / float(Kf[1, 1]), d), dim=-1).reshape(-1, 3)
    drop = depth_edge_mask(d, **edge_kw).reshape(-1) if edge_filter else None
    return points_render(pts, rgb.reshape(-1, 3), cam, K, (H, W), morph=True, drop=drop)
