"""The entry point of the hot path: `python -m worldforge_amd.infer ...` = the reference's `infer_worldforge.py` (INFER:153-339) on this engine.

    INFER:153-156  read the warped frames / masks of --video-ref            -> harness.read_frames_from_directory
    INFER:158-202  model folder, VAE fp32, pipeline bf16, clean scheduler   -> AutoencoderKLWan / WanTransformer3DModel.from_pretrained (own
                                                                               safetensors reader, checkpoint.py), UniPCMultistepScheduler.from_config
    INFER:206-254  size rule, frame / mask resize, optional mask softening  -> harness.prepare_inputs (wf_soften_mask on the device)
    INFER:256-309  prompts, seed 42, pipe(...)                              -> WanImageToVideoPipeline.__call__ (every tensor op in libwf_hip.so)
    INFER:311-339  export                                                   -> harness.save_png_frames (the mp4 container is an external encoder)

Same argument names, defaults and meaning as the reference's CLI.  Outside SURVEY section 8 and therefore NOT re-implemented: the UMT5 text encoder
and the CLIP vision encoder (run once per video; the loop consumes their outputs).  Their outputs come in through --embeds (a .npz / .safetensors
with `prompt_embeds` [1,L,4096], `negative_prompt_embeds`, `image_embeds` [1,257,1280]) or, when the checkpoint folder holds `text_encoder/`,
`tokenizer/`, `image_encoder/`, `image_processor/` and `transformers` is importable, are computed with those classes exactly as PIPE:166-214 does.
The reference's scene -> prompt table (utils/prompts.py) is text data of the reference and is not shipped: pass --prompt; the negative
prompt defaults to the entry point's own two literals (INFER:277-285, by --static).
"""
from __future__ import annotations

import argparse
import json
import os
from typing import Dict, Optional

import numpy as np
import torch

from . import harness


# INFER:277-278: the entry point's two negative prompts (protocol constants of the CLI, selected by --static at INFER:280-285)
NEGATIVE_PROMPT_STATIC = "Blink, twinkle, waggle, speak, wind, windy, leaves shaking, leaves tremble, sighboard, background dynamics, dynamic imagery, gray sky, hazy sky, overcast, gloomy sky, dim, murky, smoggy, shake, object motion blur, streaking objects, object jitter, camera shake, time flow, illogical composition, bright tones, overexposed, blurred details, subtitles, text, logo, overall gray, worst quality, low quality, JPEG compression residue, ugly, incomplete, sudden scene shift, incoherent scene jump, extra fingers, poorly drawn hands, poorly drawn faces, deformed, disfigured, misshapen limbs, fused fingers, any movement, character motion, slight object movement, object swaying, character micro-movements, subtle object rotation, object vibration, messy background, three legs, many people in the background, walking, scene changes, visual detail movement, object disintegration, object breakage."
NEGATIVE_PROMPT_DYNAMIC = "Streaking objects, mosaic, grainy, pixelated, noise, flickering, cropped, glitch, fragmented, broken, artifacts, chromatic aberration, micro camera shake, grid, tiling, blurry, camera shake, sudden scene shift, incoherent scene jump, sudden object appearance, blinking, object jitter, camera shake, illogical composition, bright tones, overexposed, blurred details, subtitles, overall gray, solid color, worst quality, low quality, JPEG compression residue, ugly, incomplete, extra fingers, poorly drawn hands, poorly drawn faces, deformed, disfigured, misshapen limbs, fused fingers, messy background, three legs, many people in the background, walking backwards"


def load_embeds(path: str, device) -> Dict[str, torch.Tensor]:
    """prompt_embeds / negative_prompt_embeds / image_embeds from a .npz or .safetensors file -> bf16 device tensors."""
    if path.endswith(".npz"):
        z = np.load(path)
        d = {k: torch.from_numpy(np.asarray(z[k], dtype=np.float32)) for k in z.files}
    else:
        from .checkpoint import load_file
        d = {k: v.float() for k, v in load_file(path).items()}
    need = ("prompt_embeds", "negative_prompt_embeds", "image_embeds")
    missing = [k for k in need if k not in d]
    if missing:
        raise ValueError(f"{path}: missing {missing} (need {need})")
    return {k: d[k].to(torch.bfloat16).to(device) for k in need}


def encode_with_transformers(model_path: str, prompt: str, negative_prompt: str, image, device, max_sequence_length: int = 512):
    """PIPE:166-214 with the Hugging Face classes the reference itself uses (outside the hot path, once per video): UMT5 last_hidden_state
    truncated to each prompt's length and zero-padded to 512 rows (PIPE:190-199); CLIP vision hidden_states[-2] (PIPE:209-211)."""
    from transformers import AutoTokenizer, CLIPImageProcessor, CLIPVisionModel, UMT5EncoderModel
    tok = AutoTokenizer.from_pretrained(os.path.join(model_path, "tokenizer"), local_files_only=True)
    te = UMT5EncoderModel.from_pretrained(os.path.join(model_path, "text_encoder"), torch_dtype=torch.bfloat16, local_files_only=True).to(device)
    out = {}
    for key, text in (("prompt_embeds", prompt), ("negative_prompt_embeds", negative_prompt)):
        t = tok([text], padding="max_length", max_length=max_sequence_length, truncation=True, add_special_tokens=True,
                return_attention_mask=True, return_tensors="pt")
        n = int(t.attention_mask.gt(0).sum(dim=1)[0])
        with torch.no_grad():
            h = te(t.input_ids.to(device), t.attention_mask.to(device)).last_hidden_state.to(torch.bfloat16)[0, :n]
        out[key] = torch.cat([h, h.new_zeros(max_sequence_length - n, h.shape[1])]).unsqueeze(0)
    del te
    proc = CLIPImageProcessor.from_pretrained(os.path.join(model_path, "image_processor"), local_files_only=True)
    ie = CLIPVisionModel.from_pretrained(os.path.join(model_path, "image_encoder"), torch_dtype=torch.float32, local_files_only=True).to(device)
    with torch.no_grad():
        px = proc(images=image, return_tensors="pt").to(device)
        out["image_embeds"] = ie(**px, output_hidden_states=True).hidden_states[-2].to(torch.bfloat16)
    return out


def run(models_dir: Optional[str], video_ref: str, model: str = "720p", output: str = "output.mp4", image: Optional[str] = None,
        guided: bool = False, resample_steps: int = 3, guide_steps: int = 20, omega: float = 1.8, omega_resample: float = 1.0,
        num_frames: int = 25, num_inference_steps: int = 50, guidance_scale: float = 5.0, resample_round: int = 20, static: bool = False,
        prompt: Optional[str] = None, negative_prompt: Optional[str] = None, embeds: Optional[str] = None,
        use_pca_channel_selection: bool = False, soften_mask: bool = False, transition_distance: int = 15, decay_type: str = "sine",
        save_png: bool = False, device: str = "cuda:0", components: Optional[dict] = None, max_area: Optional[int] = None, seed: int = 42,
        vae_precision: str = "fp16x3", flow_backend: str = "farneback"):
    """INFER:153-339.  Returns (frames float32 [F,H,W,3] in [0,1], output directory of the PNG frames or None).
    components: {"transformer", "vae", "scheduler"} to use instead of loading `models_dir` (tests; synthetic weights); max_area overrides the
    model's pixel budget the same way harness.prepare_inputs documents."""
    from .dit import WanTransformer3DModel
    from .pipeline import WanImageToVideoPipeline
    from .scheduler import UniPCMultistepScheduler
    from .vae import AutoencoderKLWan

    dev = torch.device(device)
    model_path = None
    if components is None:
        model_path = os.path.join(models_dir, "wan2.1_480p_model_local" if model == "480p" else "wan2.1_720p_model_local")   # INFER:160-168
        if not os.path.exists(model_path):
            raise ValueError(f"Model path does not exist: {model_path}")
        vae = AutoencoderKLWan.from_pretrained(model_path, device=dev, precision=vae_precision)                              # INFER:185-189
        transformer = WanTransformer3DModel.from_pretrained(model_path, device=dev)                                          # INFER:191-197
        cfg_file = os.path.join(model_path, "scheduler", "scheduler_config.json")
        sconf = json.load(open(cfg_file)) if os.path.exists(cfg_file) else {"flow_shift": 3.0 if model == "480p" else 5.0}
        scheduler = UniPCMultistepScheduler.from_config(sconf, flow_backend=flow_backend)                                    # INFER:200-202
    else:
        transformer, vae, scheduler = components["transformer"], components["vae"], components["scheduler"]
    pipe = WanImageToVideoPipeline(transformer, vae, scheduler, device=dev)

    # INFER:206-254 (the first frame of the warped sequence is the input image unless --image is given)
    # --image given: the size rule takes the IMAGE's aspect ratio and the warped frames / masks are resized to it (INFER:209-241).  The
    # warped sequence is NOT truncated to --num-frames: the reference blends frame by frame and its resize of a sequence of another length
    # raises (scheduling_unipc_multistep_clean.py:1326), so a mismatch is an error here too, before any GPU work.
    pil, video, mask, height, width = harness.prepare_inputs(video_ref, model=model, num_frames=None, soften=soften_mask,
                                                             transition_distance=transition_distance, decay_type=decay_type, device=dev,
                                                             max_area=max_area, image=image)
    eff_frames = max(num_frames // 4 * 4 + 1 if num_frames % 4 != 1 else num_frames, 1)     # PIPE:475-478
    if video.shape[2] != eff_frames:
        raise ValueError(f"--video-ref holds {video.shape[2]} frames but --num-frames {num_frames} decodes {eff_frames}: the reference blends "
                         "frame by frame (scheduling_unipc_multistep_clean.py:1326 raises on a mismatch)")
    if negative_prompt is None:   # INFER:277-284: the two negative prompts are literals of the entry point, chosen by --static
        negative_prompt = NEGATIVE_PROMPT_STATIC if static else NEGATIVE_PROMPT_DYNAMIC

    if embeds is not None:
        emb = load_embeds(embeds, dev)
    elif model_path is not None and all(os.path.isdir(os.path.join(model_path, d)) for d in ("text_encoder", "tokenizer", "image_encoder", "image_processor")):
        if prompt is None:
            raise ValueError("pass --prompt (the reference's scene -> prompt table utils/prompts.py is not shipped)")
        emb = encode_with_transformers(model_path, prompt, negative_prompt, pil, dev)
    else:
        raise ValueError("no --embeds file and no text_encoder / image_encoder folders to compute them from")

    out = pipe(image=pil, height=height, width=width, num_frames=num_frames, num_inference_steps=num_inference_steps,
               guidance_scale=guidance_scale, generator=torch.manual_seed(seed), prompt_embeds=emb["prompt_embeds"],
               negative_prompt_embeds=emb["negative_prompt_embeds"], image_embeds=emb["image_embeds"], output_type="np", video_ref=video,
               mask=mask, guided=guided, resample_steps=resample_steps, guide_steps=guide_steps, omega=omega, omega_resample=omega_resample,
               resample_round=resample_round, use_pca_channel_selection=use_pca_channel_selection, static=static)
    frames = out.frames[0]
    png_dir = None
    out_dir = os.path.dirname(output)
    if out_dir:
        os.makedirs(out_dir, exist_ok=True)
    # the mp4 container (diffusers' export_to_video, INFER:317) is an external encoder, so the lossless PNG frames (--save-png in the
    # reference, INFER:323-339) are this engine's output format and are always written
    png_dir = harness.save_png_frames(frames, output)
    return frames, png_dir


def main(argv=None):
    ap = argparse.ArgumentParser(description="WorldForge guided image-to-video on MI355X (the reference's infer_worldforge.py arguments)")
    ap.add_argument("--model", choices=["480p", "720p"], default="720p")
    ap.add_argument("--models-dir", required=True)
    ap.add_argument("--output", default="output.mp4")
    ap.add_argument("--image", default=None)
    ap.add_argument("--video-ref", required=True)
    ap.add_argument("--guided", action="store_true")
    ap.add_argument("--resample-steps", type=int, default=3)
    ap.add_argument("--guide-steps", type=int, default=20)
    ap.add_argument("--omega", type=float, default=1.8)
    ap.add_argument("--omega_resample", type=float, default=1.0)
    ap.add_argument("--num-frames", type=int, default=25)
    ap.add_argument("--num-inference-steps", type=int, default=50)
    ap.add_argument("--guidance-scale", type=float, default=5.0)
    ap.add_argument("--resample-round", type=int, default=20)
    ap.add_argument("--static", choices=["True", "False"], default="False")
    ap.add_argument("--use-pca-channel-selection", action="store_true")
    ap.add_argument("--soften-mask", action="store_true")
    ap.add_argument("--transition-distance", type=int, default=15)
    ap.add_argument("--decay-type", choices=["linear", "exponential", "sine", "cosine"], default="sine")
    ap.add_argument("--save-png", action="store_true",
                    help="accepted for CLI compatibility: the lossless PNG frames are ALWAYS written (this engine has no mp4 encoder, INFER:317)")
    ap.add_argument("--prompt", default=None, help="instead of the reference's --scene lookup in utils/prompts.py")
    ap.add_argument("--negative-prompt", default=None, help="default: the reference's static / dynamic negative prompt by --static (INFER:277-285)")
    ap.add_argument("--embeds", default=None, help=".npz / .safetensors with prompt_embeds, negative_prompt_embeds, image_embeds")
    ap.add_argument("--device", default="cuda:0")
    a = ap.parse_args(argv)
    frames, png_dir = run(a.models_dir, a.video_ref, model=a.model, output=a.output, image=a.image, guided=a.guided,
                          resample_steps=a.resample_steps, guide_steps=a.guide_steps, omega=a.omega, omega_resample=a.omega_resample,
                          num_frames=a.num_frames, num_inference_steps=a.num_inference_steps, guidance_scale=a.guidance_scale,
                          resample_round=a.resample_round, static=a.static == "True", prompt=a.prompt, negative_prompt=a.negative_prompt,
                          embeds=a.embeds, use_pca_channel_selection=a.use_pca_channel_selection, soften_mask=a.soften_mask,
                          transition_distance=a.transition_distance, decay_type=a.decay_type, save_png=a.save_png, device=a.device)
    print(f"{len(frames)} frames -> {png_dir}")


if __name__ == "__main__":
    main()
