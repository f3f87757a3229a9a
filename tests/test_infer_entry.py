"""`worldforge_amd.infer` = the reference's entry point infer_worldforge.py:153-339 on this engine.  CPU: argument surface, embedding
files, error behaviour.  GPU: `infer.run()` on the truck fixture (BASELINE configs[0]'s inputs) with injected small components writes
the PNG frames and equals a direct pipeline call."""
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TRUCK = os.path.join(ROOT, "tests", "golden", "truck")


def test_cli_has_the_reference_arguments_and_defaults():
    """INFER:19-63: same flags, same defaults (the --scene lookup is replaced by --prompt / --embeds)."""
    from worldforge_amd import infer
    import argparse
    ap_args = {}
    real = argparse.ArgumentParser.parse_args

    def grab(self, argv=None):
        ns = real(self, argv)
        ap_args.update(vars(ns))
        raise SystemExit(0)

    argparse.ArgumentParser.parse_args = grab
    try:
        with pytest.raises(SystemExit):
            infer.main(["--models-dir", "/nowhere", "--video-ref", TRUCK])
    finally:
        argparse.ArgumentParser.parse_args = real
    want = dict(model="720p", output="output.mp4", image=None, guided=False, resample_steps=3, guide_steps=20, omega=1.8, omega_resample=1.0,
                num_frames=25, num_inference_steps=50, guidance_scale=5.0, resample_round=20, static="False",
                use_pca_channel_selection=False, soften_mask=False, transition_distance=15, decay_type="sine", save_png=False)
    for k, v in want.items():
        assert ap_args[k] == v, (k, ap_args[k], v)


def test_embeds_file_round_trip_and_missing_keys(tmp_path):
    from worldforge_amd import infer
    p = tmp_path / "e.npz"
    np.savez(p, prompt_embeds=np.ones((1, 512, 8), np.float32), negative_prompt_embeds=np.zeros((1, 512, 8), np.float32),
             image_embeds=np.full((1, 257, 4), 0.5, np.float32))
    e = infer.load_embeds(str(p), "cpu")
    assert e["prompt_embeds"].dtype == torch.bfloat16 and tuple(e["image_embeds"].shape) == (1, 257, 4)
    q = tmp_path / "bad.npz"
    np.savez(q, prompt_embeds=np.ones((1, 2, 2), np.float32))
    with pytest.raises(ValueError, match="missing"):
        infer.load_embeds(str(q), "cpu")


def test_missing_model_folder_raises_like_the_reference(tmp_path):
    from worldforge_amd import infer
    with pytest.raises(ValueError, match="Model path does not exist"):      # INFER:170-171
        infer.run(str(tmp_path), TRUCK, model="480p")


@pytest.mark.gpu
def test_run_on_the_truck_fixture_writes_frames_and_equals_the_direct_pipeline_call(tmp_path):
    from oracle import dit as odit
    from oracle import vae as ovae
    from worldforge_amd import dit, harness, infer
    from worldforge_amd.pipeline import WanImageToVideoPipeline
    from worldforge_amd.scheduler import UniPCMultistepScheduler
    from worldforge_amd.vae import AutoencoderKLWan
    dev = torch.device("cuda:0")
    ocfg = odit.DiTConfig(dim=256, ffn_dim=512, num_heads=2, num_layers=2, text_dim=64)
    model = dit.WanTransformer3DModel(dit.DiTConfig(dim=256, ffn_dim=512, num_heads=2, num_layers=2, text_dim=64), dev).load_state_dict(
        odit.random_weights(ocfg, seed=3))
    vae = AutoencoderKLWan(dev).load_state_dict(ovae.random_weights(seed=4))
    g = torch.Generator().manual_seed(1)
    emb = tmp_path / "embeds.npz"
    np.savez(emb, prompt_embeds=(torch.randn(1, 24, 64, generator=g) * 0.5).numpy(), negative_prompt_embeds=(torch.randn(1, 24, 64, generator=g) * 0.5).numpy(),
             image_embeds=torch.randn(1, 257, 1280, generator=g).numpy())
    kw = dict(guided=True, resample_steps=2, guide_steps=3, resample_round=3, omega=4.0, omega_resample=4.0, num_frames=9,
              num_inference_steps=4, guidance_scale=4.0, use_pca_channel_selection=True, soften_mask=True, static=True)
    comps = lambda: dict(transformer=model, vae=vae, scheduler=UniPCMultistepScheduler(flow_shift=3.0))  # noqa: E731
    frames, png_dir = infer.run(None, TRUCK, model="480p", output=str(tmp_path / "out" / "truck.mp4"), embeds=str(emb), components=comps(),
                                max_area=64 * 112, device="cuda:0", **kw)
    assert frames.shape == (9, 48, 112, 3) and np.isfinite(frames).all()
    files = sorted(os.listdir(png_dir))
    assert files == [f"frame_{i:04d}.png" for i in range(9)] and png_dir.endswith("truck_frames")
    from PIL import Image
    back = np.asarray(Image.open(os.path.join(png_dir, files[4])))
    assert np.array_equal(back, (frames[4] * 255).clip(0, 255).astype(np.uint8))                  # INFER:331-333
    # the same job wired by hand
    image, ref, mask, h, w = harness.prepare_inputs(TRUCK, model="480p", num_frames=9, soften=True, device=dev, max_area=64 * 112)
    e = infer.load_embeds(str(emb), dev)
    pipe = WanImageToVideoPipeline(model, vae, UniPCMultistepScheduler(flow_shift=3.0), device=dev)
    out = pipe(image=image, height=h, width=w, num_frames=9, num_inference_steps=4, guidance_scale=4.0, generator=torch.manual_seed(42),
               output_type="np", video_ref=ref, mask=mask, guided=True, resample_steps=2, guide_steps=3, omega=4.0, omega_resample=4.0,
               resample_round=3, use_pca_channel_selection=True, static=True, **e)
    assert np.array_equal(out.frames[0], frames)
    with pytest.raises(ValueError, match="frames"):                                                  # SCHED:1326 surfaced early
        infer.run(None, TRUCK, model="480p", embeds=str(emb), components=comps(), max_area=64 * 112, **{**kw, "num_frames": 5 + 8})
    with pytest.raises(ValueError, match="frames"):        # a LONGER warped sequence is not silently truncated either
        infer.run(None, TRUCK, model="480p", embeds=str(emb), components=comps(), max_area=64 * 112, **{**kw, "num_frames": 5})
