"""CPU: oracle/warp.py against the unmodified reference warp_single_img (fill_cracks=False), tests/golden/g16_warp.npz."""
import os

import numpy as np

from oracle import warp as owarp

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "g16_warp.npz"))


def test_splat_equals_reference():
    for name in ("right", "forward"):
        cams = list(G[f"{name}_cams"])
        imgs, masks, depths = owarp.splat(G["image"], G["depth"], G["K"], G["E"], cams[1:])
        assert np.array_equal(imgs, G[f"{name}_imgs"][1:]) and np.array_equal(masks, G[f"{name}_masks"][1:])
        # frame 0 of the reference is the untouched original
        assert np.array_equal(G[f"{name}_imgs"][0], (G["image"] * 255).astype(np.uint8))
        assert np.isnan(depths[masks == 0]).all() and np.isfinite(depths[masks == 1]).all()
        assert 0.3 < masks.mean() < 0.95  # holes from disocclusion and the invalid-depth patch


def test_camera_paths_equal_reference():
    """worldforge_amd.warp.camera_path (host-only float64 math, importable without a GPU) against the reference's ten generators."""
    import pytest
    from worldforge_amd.warp import camera_path
    C = np.load(os.path.join(os.path.dirname(__file__), "golden", "g16b_warp_cams.npz"))
    for name in ("up", "down", "right", "left", "forward", "backward", "up_pan", "down_pan", "left_pan", "right_pan"):
        got = np.stack(camera_path(name, C["E"], 17.0, 6, 2.3))
        assert got.shape == C[name].shape
        assert np.abs(got - C[name]).max() <= 1e-12, (name, np.abs(got - C[name]).max())
    with pytest.raises(ValueError):
        camera_path("sideways", C["E"], 1.0, 2, 1.0)
