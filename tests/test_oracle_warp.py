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
