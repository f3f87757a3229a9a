"""Record golden vectors from the THIRD-PARTY packages the reference calls on the hot path but which exist neither in /root/reference nor in
this image: opencv-python (`cv2`) and pytorch3d.  Run it once on any machine that has them; it needs nothing else but numpy (+ scipy for
one input) and this repo's tests/ package:

    pip install opencv-python            # requirements.txt:8 of the reference (unpinned)
    python tools/record_thirdparty_goldens.py                # writes tests/golden/g20_farneback_cv2.npz, g21_crackfill_cv2.npz
    pip install pytorch3d                # DepthCrafter/requirements: the point rasteriser
    python tools/record_thirdparty_goldens.py pointrender    # writes tests/golden/g22_pointrender_pytorch3d.npz

Every INPUT is generated from a seed by tests/thirdparty_cases.py and stored next to the outputs (uint8 frames, masks, points: libm differences
between machines could otherwise move a pixel across a quantisation boundary); the files stay under a megabyte.
As soon as a file exists, tests/test_thirdparty_goldens.py (CPU: oracle/ vs the package) and tests/test_gpu_thirdparty_goldens.py (GPU: the
HIP kernels vs the package) stop skipping, and the "parity unpinned" labels of oracle/farneback.py, oracle/crackfill.py and
oracle/pointrender.py can be dropped.  What is recorded, with the reference call site each vector pins:

  g20  cv2.calcOpticalFlowFarneback(prev, next, None, 0.5, 3, 15, 3, 5, 1.2, 0)      utils/scheduling_unipc_multistep_clean.py:220-224
       on uint8 frames prepared as :165-201 prepares them (global-range normalisation, x255, truncation; RGB2GRAY of 3 equal channels),
       for a latent-sized clip (16 channels x 5 frames x 60 x 104) and a small odd-sized one; plus cv2.cvtColor(RGB2GRAY) of that input;
  g21  cv2.filter2D(img, -1, ones(3,3)) with the default border and cv2.morphologyEx(mask, MORPH_CLOSE, ones(3,3))
       vggt/modules/utils_warp.py:386-430 (the two OpenCV primitives of the crack filling), on a seeded image / hole mask;
  g22  pytorch3d PointsRasterizer(radius 0.005, points_per_pixel 1) through cameras_from_opencv_projection, cv2.morphologyEx(MORPH_OPEN, 5x5)
       DepthCrafter/utils.py:103-171, and cv2.Sobel / cv2.dilate of :495-560, on a seeded depth map.
"""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden")

from tests import thirdparty_cases as tc  # noqa: E402


def record_farneback(cv2, out_dir=OUT):
    out = {"cv2_version": np.array(cv2.__version__)}
    for name in tc.FARNEBACK_CASES:
        frames = tc.farneback_frames(name)                    # uint8 [C, T, h, w]
        C, T = frames.shape[:2]
        flows = np.zeros((C, T - 1, 2) + frames.shape[2:], dtype=np.float32)
        for c in range(C):
            for t in range(T - 1):
                rgb1, rgb2 = (np.repeat(frames[c, t + d][..., None], 3, axis=2) for d in (0, 1))
                g1, g2 = cv2.cvtColor(rgb1, cv2.COLOR_RGB2GRAY), cv2.cvtColor(rgb2, cv2.COLOR_RGB2GRAY)
                assert (g1 == frames[c, t]).all()              # luma of three equal channels is the identity
                fl = cv2.calcOpticalFlowFarneback(g1, g2, None, pyr_scale=0.5, levels=3, winsize=15, iterations=3, poly_n=5,
                                                  poly_sigma=1.2, flags=0)
                flows[c, t] = fl.transpose(2, 0, 1)
        out[f"{name}_flows"] = flows
        out[f"{name}_frames"] = frames
    np.savez_compressed(os.path.join(out_dir, "g20_farneback_cv2.npz"), **out)
    print("g20_farneback_cv2", {k: v.shape for k, v in out.items() if k.endswith("_flows")})


def record_crackfill(cv2, out_dir=OUT):
    img, mask = tc.crackfill_inputs()
    k = np.ones((3, 3), np.float32)
    out = {"cv2_version": np.array(cv2.__version__), "img": img, "mask": mask,
           "filter2d": np.stack([cv2.filter2D(img[..., c], -1, k) for c in range(3)], axis=-1),
           "filter2d_mask": cv2.filter2D(mask.astype(np.float32), -1, k),
           "close3": cv2.morphologyEx(mask, cv2.MORPH_CLOSE, np.ones((3, 3), np.uint8))}
    np.savez_compressed(os.path.join(out_dir, "g21_crackfill_cv2.npz"), **out)
    print("g21_crackfill_cv2", {k: v.shape for k, v in out.items() if k != "cv2_version"})


def record_pointrender(out_dir=OUT):
    import cv2
    import torch
    from pytorch3d.renderer import PointsRasterizationSettings, PointsRasterizer
    from pytorch3d.structures import Pointclouds
    from pytorch3d.utils import cameras_from_opencv_projection

    pts, ext, K, (H, W), depth = tc.pointrender_inputs()
    cams = cameras_from_opencv_projection(torch.from_numpy(ext[None, :3, :3]), torch.from_numpy(ext[None, :3, 3]),
                                          torch.from_numpy(K[None]), torch.tensor([[H, W]], dtype=torch.float32))
    rast = PointsRasterizer(cameras=cams, raster_settings=PointsRasterizationSettings(image_size=(H, W), radius=0.005, points_per_pixel=1))
    frag = rast(Pointclouds(points=[torch.from_numpy(pts)]))
    idx = frag.idx[0, :, :, 0].numpy().astype(np.int64)
    cover = (idx >= 0).astype(np.uint8)
    gx, gy = cv2.Sobel(depth, cv2.CV_64F, 1, 0, ksize=3), cv2.Sobel(depth, cv2.CV_64F, 0, 1, ksize=3)
    out = {"points": pts, "extrinsic": ext, "K": K, "size_hw": np.array([H, W]), "depth": depth, "idx": idx, "open5": cv2.morphologyEx(cover, cv2.MORPH_OPEN, np.ones((5, 5), np.uint8)),
           "sobel_mag": np.sqrt(gx ** 2 + gy ** 2), "dilate7": cv2.dilate(cover, np.ones((7, 7), np.uint8), iterations=1)}
    np.savez_compressed(os.path.join(out_dir, "g22_pointrender_pytorch3d.npz"), **out)
    print("g22_pointrender_pytorch3d", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    which = sys.argv[1:] or ["farneback", "crackfill"]
    if "farneback" in which or "crackfill" in which:
        import cv2
        if "farneback" in which:
            record_farneback(cv2)
        if "crackfill" in which:
            record_crackfill(cv2)
    if "pointrender" in which:
        record_pointrender()
