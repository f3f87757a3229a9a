"""CPU: front-end helpers (mask softening, size rule, directory reader) against goldens from the reference functions."""
import os

import numpy as np
import pytest

from oracle import harness as oh
from worldforge_amd import harness as ph


def test_soften_mask_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "g10_harness.npz"))
    masks = g["masks"]
    for decay in ("linear", "exponential", "sine", "cosine"):
        for d in (5, 15):
            want = g[f"soft_{decay}_{d}"]
            np.testing.assert_array_equal(oh.soften_mask(masks, d, decay), want)
            np.testing.assert_array_equal(ph.soften_mask(masks, d, decay), want)
    with pytest.raises(ValueError):
        ph.soften_mask(masks, 5, "bogus")


def test_size_rule_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "g10_harness.npz"))
    for ih, iw, area, h, w in g["size_rule"].tolist():
        assert oh.size_rule(ih, iw, area) == (h, w)
        assert ph.target_size(ih, iw, area) == (h, w)
    assert ph.target_size(720, 1280, 480 * 832) == (464, 832)  # test_case/truck at 480p (C1)


def test_read_frames_from_directory(tmp_path):
    from PIL import Image
    for i in range(3):
        Image.new("RGB", (8, 6), (i * 40, 0, 0)).save(tmp_path / f"warp_{i:02d}.png")
    for i in range(2):
        Image.new("L", (8, 6), 255).save(tmp_path / f"mask_{i:02d}.png")
    frames, masks, first = ph.read_frames_from_directory(str(tmp_path))
    assert len(frames) == 3 and len(masks) == 3 and first.size == (8, 6)   # masks padded with the last one (INFER:95-99)
    assert frames[1].getpixel((0, 0))[0] == 40
    with pytest.raises(ValueError):
        ph.read_frames_from_directory(str(tmp_path / "missing"))
    img, ref, mask, h, w = ph.prepare_inputs(str(tmp_path), num_frames=1, soften=False)
    assert ref.shape == (1, 3, 1, h, w) and mask.shape == (1, 1, 1, h, w) and ref.dtype.is_floating_point


def test_save_png_frames_round_trip(tmp_path):
    """INFER:323-339: float frames are truncated to uint8 exactly as `(x * 255).clip(0, 255).astype(np.uint8)`, names frame_%04d.png."""
    from PIL import Image
    rng = np.random.default_rng(0)
    frames = rng.random((3, 6, 8, 3)).astype(np.float32)
    frames[0, 0, 0] = [1.0, 0.0, 0.99999]
    d = ph.save_png_frames(frames, str(tmp_path / "out" / "video.mp4"))
    assert d.endswith("video_frames") and sorted(os.listdir(d)) == ["frame_0000.png", "frame_0001.png", "frame_0002.png"]
    for i in range(3):
        got = np.array(Image.open(os.path.join(d, f"frame_{i:04d}.png")))
        np.testing.assert_array_equal(got, (frames[i] * 255).clip(0, 255).astype(np.uint8))
    assert tuple(np.array(Image.open(os.path.join(d, "frame_0000.png")))[0, 0]) == (255, 0, 254)
