"""Tracing: roctx ranges + per-phase GPU time log of the sampling loop."""
import json

import pytest
import torch


def test_disabled_tracer_is_a_no_op():
    from worldforge_amd import trace
    t = trace.Tracer(enabled=False)
    with t.range("x", step=1):
        pass
    assert t.resolve() == [] and t.summary() == {} and not trace.NULL.enabled
    assert not trace.Tracer(enabled=True).enabled or torch.cuda.is_available()   # needs a GPU to time anything


@pytest.mark.gpu
def test_guided_job_trace_has_every_phase_with_the_reference_call_counts(tmp_path):
    """4 steps, guide = round = 3, R = 2 (the shape of BASELINE config 1): 3 guided steps x 2 rounds + 1 plain step -> 7 CFG
    evaluations, 7 scheduler steps, 6 decode / blend / encode, 3 FLF gates (round 0 only, SCHED:1391), 3 re-noise, 3 DSG, 1 final
    decode -- the counts SURVEY 3.1 derives from the reference loop."""
    from worldforge_amd import dit, trace
    from worldforge_amd.pipeline import WanImageToVideoPipeline
    from worldforge_amd.scheduler import UniPCMultistepScheduler
    from worldforge_amd.vae import AutoencoderKLWan
    dev = torch.device("cuda:0")
    cfg = dit.DiTConfig(dim=256, ffn_dim=512, num_heads=2, num_layers=2, text_dim=64)
    g = torch.Generator().manual_seed(3)
    Fr, H, Wd = 9, 32, 48
    image = torch.rand(3, H, Wd, generator=g)
    ref = torch.rand(1, 3, Fr, H, Wd, generator=g)
    mask = (torch.rand(1, 1, Fr, H, Wd, generator=g) > 0.3).float()
    text, neg = torch.randn(1, 30, 64, generator=g).bfloat16(), torch.randn(1, 30, 64, generator=g).bfloat16()
    img = torch.randn(1, 257, 1280, generator=g).bfloat16()
    pipe = WanImageToVideoPipeline(dit.WanTransformer3DModel(cfg, dev).init_random(5), AutoencoderKLWan(dev).init_random(seed=1),
                                   UniPCMultistepScheduler(flow_shift=3.0), device=dev)
    pipe.tracer = trace.Tracer(enabled=True)
    pipe.tracer.path = str(tmp_path / "timing.json")
    pipe(image=image, height=H, width=Wd, num_frames=Fr, num_inference_steps=4, guidance_scale=4.0, generator=torch.manual_seed(1),
         prompt_embeds=text, negative_prompt_embeds=neg, image_embeds=img, output_type="np", video_ref=ref, mask=mask, static=True,
         guided=True, resample_steps=2, guide_steps=3, omega=4.0, omega_resample=4.0, resample_round=3, use_pca_channel_selection=True)
    counts = {k: v["count"] for k, v in pipe.timing.items()}
    assert counts == {"dit_cfg": 7, "scheduler_step": 7, "vae_decode": 6, "blend": 6, "vae_encode": 6, "flf_gate": 3, "renoise": 3,
                      "dsg": 3, "final_decode": 1}, counts
    assert all(v["total_ms"] > 0 for v in pipe.timing.values())
    log = json.load(open(tmp_path / "timing.json"))
    assert len(log["records"]) == sum(counts.values()) and {"name", "ms"} <= set(log["records"][0])
    assert any(r["name"] == "vae_decode" and r.get("step") == 2 for r in log["records"])
