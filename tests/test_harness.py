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


def test_prepare_inputs_takes_the_size_from_the_input_image(tmp_path):
    """INFER:209-241: with --image the size rule uses the IMAGE's aspect ratio and the warped frames / masks are resized to it."""
    from PIL import Image
    seq = tmp_path / "imgs"
    seq.mkdir()
    for i in range(2):
        Image.new("RGB", (64, 32), (i * 40, 9, 0)).save(seq / f"warp_{i:02d}.png")
        Image.new("L", (64, 32), 255).save(seq / f"mask_{i:02d}.png")
    square = tmp_path / "input.png"
    Image.new("RGB", (40, 40), (1, 2, 3)).save(square)
    _, ref0, _, h0, w0 = ph.prepare_inputs(str(seq), soften=False, max_area=64 * 32)          # no --image: the first frame's 1 : 2
    img, ref, mask, h, w = ph.prepare_inputs(str(seq), soften=False, max_area=64 * 32, image=str(square))
    assert (h0, w0) == ph.target_size(32, 64, 64 * 32) and (h, w) == ph.target_size(40, 40, 64 * 32) and h == w and (h, w) != (h0, w0)
    assert img.size == (w, h) and img.getpixel((0, 0)) == (1, 2, 3)
    assert ref.shape == (1, 3, 2, h, w) and mask.shape == (1, 1, 2, h, w)


def test_infer_negative_prompt_defaults_follow_static():
    """INFER:277-285: the two negative prompts are literals of the entry point, picked by --static."""
    from worldforge_amd import infer
    assert infer.NEGATIVE_PROMPT_STATIC.startswith("Blink, twinkle") and infer.NEGATIVE_PROMPT_DYNAMIC.startswith("Streaking objects")
    if os.path.isdir("/root/reference/wan_for_worldforge"):
        src = open("/root/reference/wan_for_worldforge/infer_worldforge.py").read()
        assert infer.NEGATIVE_PROMPT_STATIC in src and infer.NEGATIVE_PROMPT_DYNAMIC in src


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


def test_align_reference_memo_returns_the_same_tensors_while_the_inputs_are_unchanged():
    """a float64 mask (what INFER:242-247 hands over when the mask is not softened) is cast on every injection; with the
    scheduler's memo the cast result is reused -- so the sharded VAE's row-slab cache, keyed on the tensors it is handed, hits -- and an
    in-place change of the source invalidates it."""
    import torch
    from worldforge_amd.scheduler import align_reference
    ref, mask = torch.rand(1, 3, 5, 16, 16), torch.rand(1, 1, 5, 16, 16, dtype=torch.float64)
    memo = {}
    a = align_reference(ref, mask, (1, 3, 5, 16, 16), memo=memo)
    b = align_reference(ref, mask, (1, 3, 5, 16, 16), memo=memo)
    assert a[0] is ref and a[1].dtype == torch.float32 and b[1] is a[1] and len(memo) == 1
    mask.mul_(0.5)
    c = align_reference(ref, mask, (1, 3, 5, 16, 16), memo=memo)
    assert c[1] is not a[1] and torch.equal(c[1], mask.float()) and len(memo) == 1
    assert align_reference(ref, mask, (1, 3, 5, 16, 16))[1] is not c[1]          # no memo: a fresh cast every call, as before
