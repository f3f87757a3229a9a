"""Front-end of the hot path: the host-side pre-processing `infer_worldforge.py` performs once per video (INFER:65-150,
217-254).  Pure host code (PIL / numpy / scipy), as in the reference: it runs once, outside the sampling loop."""
from __future__ import annotations

import glob
import os
from typing import List, Tuple

import numpy as np
import torch


def read_frames_from_directory(directory: str):
    """INFER:65-102: sorted images of a directory; files starting with 'mask_' are masks (padded / truncated to #frames)."""
    from PIL import Image

    files: List[str] = []
    for ext in ("*.jpg", "*.jpeg", "*.png", "*.bmp", "*.tiff"):
        files.extend(glob.glob(os.path.join(directory, ext)))
    files = sorted(files)
    if not files:
        raise ValueError(f"No image files found in directory {directory}")
    frame_files = [f for f in files if not os.path.basename(f).startswith("mask_")]
    mask_files = [f for f in files if os.path.basename(f).startswith("mask_")]
    frames = [Image.open(f).convert("RGB") for f in frame_files]
    masks = [Image.open(f).convert("L") for f in mask_files]
    if not masks and frames:
        masks = [Image.new("L", frames[0].size, 0) for _ in frames]
    if len(masks) != len(frames):
        while len(masks) < len(frames):
            masks.append(masks[-1] if masks else Image.new("L", frames[0].size, 0))
        masks = masks[:len(frames)]
    return frames, masks, (frames[0] if frames else None)


def soften_mask(mask_array: np.ndarray, transition_distance: int = 15, decay_type: str = "sine") -> np.ndarray:
    """INFER:105-150: inside the ones-region, pixels within `transition_distance` of the boundary get a smooth ramp of
    their Euclidean distance to the zero-region."""
    from scipy.ndimage import distance_transform_edt

    out = mask_array.copy().astype(np.float32)
    for i in range(mask_array.shape[0]):
        cur = mask_array[i].astype(bool)
        if np.all(cur) or np.all(~cur):
            continue
        frame = mask_array[i].copy().astype(np.float32)
        dist = distance_transform_edt(cur)
        ramp = cur & (dist <= transition_distance)
        if np.any(ramp):
            t = np.clip(dist[ramp] / transition_distance, 0.0, 1.0)
            if decay_type == "linear":
                v = t
            elif decay_type == "exponential":
                v = 1.0 - np.exp(-3.0 * t)
            elif decay_type == "sine":
                v = np.sin(np.pi / 2 * t)
            elif decay_type == "cosine":
                v = 1.0 - np.cos(np.pi / 2 * t)
            else:
                raise ValueError(f"Unsupported decay type: {decay_type}")
            frame[ramp] = v
        out[i] = frame
    return out


def target_size(image_height: int, image_width: int, max_area: int, mod_value: int = 16) -> Tuple[int, int]:
    """INFER:218-221."""
    ar = image_height / image_width
    h = round(np.sqrt(max_area * ar)) // mod_value * mod_value
    w = round(np.sqrt(max_area / ar)) // mod_value * mod_value
    return int(h), int(w)


def prepare_inputs(directory: str, model: str = "480p", num_frames: int = None, soften: bool = True,
                   transition_distance: int = 15, decay_type: str = "sine", device=None, max_area: int = None, image=None):
    """INFER:153-254 -> (image PIL, video_ref [1,3,F,H,W] f32 in [0,1], mask [1,1,F,H,W], height, width).
    `num_frames` (not in the reference) truncates the warped sequence: the reference requires #reference frames == the
    (4k+1) frame count it decodes, otherwise SCHED:1326 raises.  `max_area` (not in the reference) overrides the model's pixel budget
    (INFER:217: 480*832 or 720*1280) so that the same size rule can be exercised at test sizes.  With `device` (a GPU) the mask softening runs there
    (wf_soften_mask: exact windowed EDT) and video_ref / mask are returned on that device.  `image` (a PIL image or a path; INFER:209-215
    `--image`): the input image -- the size rule takes ITS aspect ratio and the warped frames / masks are resized to that size (INFER:218-
    241); without it the first warped frame is the image."""
    frames, masks, first = read_frames_from_directory(directory)
    if image is not None:
        if isinstance(image, (str, os.PathLike)):
            from PIL import Image
            image = Image.open(image).convert("RGB")
        first = image
    if first is None:
        raise ValueError("Cannot get first frame as input image, please specify --image parameter")   # INFER:214-215
    if num_frames is not None:
        frames, masks = frames[:num_frames], masks[:num_frames]
    if max_area is None:
        max_area = 480 * 832 if model == "480p" else 720 * 1280
    h, w = target_size(first.height, first.width, max_area)
    image = first.resize((w, h))
    video = torch.stack([torch.tensor(np.array(f.resize((w, h)))).permute(2, 0, 1).float() / 255.0 for f in frames])
    video_ref = video.unsqueeze(0).permute(0, 2, 1, 3, 4)
    marr = np.stack([np.array(m.resize((w, h))) / 255.0 for m in masks])
    if device is not None and torch.device(device).type == "cuda":
        from . import ops
        m32 = torch.from_numpy(marr.astype(np.float32)).to(device)
        if soften:
            m32 = ops.soften_mask(m32.contiguous(), transition_distance, decay_type)
        return image, video_ref.to(device), m32.unsqueeze(0).unsqueeze(0), h, w
    if soften:
        marr = soften_mask(marr, transition_distance, decay_type)
    mask = torch.from_numpy(marr).unsqueeze(0).unsqueeze(0)
    return image, video_ref, mask, h, w


def save_png_frames(frames, output_path: str) -> str:
    """INFER:323-339 (`--save-png`): frames [F,H,W,3] (float in [0,1] or uint8, numpy / torch / a list of PIL images) -> PNG files
    `<output stem>_frames/frame_0000.png ...`; float frames are quantised as the reference does, `(x * 255).clip(0, 255).astype(uint8)`
    (truncation).  Returns the directory.  (The mp4 container itself -- diffusers' `export_to_video`, INFER:317 -- is an external encoder.)"""
    from PIL import Image

    png_dir = os.path.splitext(output_path)[0] + "_frames"
    os.makedirs(png_dir, exist_ok=True)
    if isinstance(frames, torch.Tensor):
        frames = frames.detach().cpu().numpy()
    for i, frame in enumerate(frames):
        if isinstance(frame, Image.Image):
            img = frame
        else:
            arr = np.array(frame)
            if arr.dtype in (np.float32, np.float64):
                arr = (arr * 255).clip(0, 255).astype(np.uint8)
            img = Image.fromarray(arr)
        img.save(os.path.join(png_dir, f"frame_{i:04d}.png"), format="PNG")
    return png_dir
